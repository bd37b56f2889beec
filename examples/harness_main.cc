// harness_main.cc — a host in the shape of the reference's harness executable
// (/root/reference/src/main.cc:11-105): define the MATRIX_* globals, fill a SampleData with host
// buffers read from a seven-file data set, call QbitGemv.  The reference's own main.cc links against
// libxbitops_harness.so unchanged (tests/test_harness_format.py builds it where /root/reference is
// mounted); this program exists because that file hard-codes its author's data directory
// (main.cc:14) and a K override (main.cc:21), so it cannot be pointed at another data set.
//
//   g++ -O2 -I include examples/harness_main.cc -o examples/harness_main -L xbitops_amd/lib \
//       -lxbitops_harness -lxbitops_hip -Wl,-rpath,'$ORIGIN/../xbitops_amd/lib'
//   examples/harness_main <directory>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "xbitops_harness.h"

int MATRIX_M = 0;  // read by the library, as the reference's kernels files do (unpack_weight_2_to_7.cu:30-32)
int MATRIX_K = 0;
int MATRIX_N = 0;

static bool read_file(const std::string& path, std::vector<char>* out, size_t bytes) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path.c_str());
    return false;
  }
  out->resize(bytes);
  const size_t got = fread(out->data(), 1, bytes, f);
  fclose(f);
  if (got != bytes) fprintf(stderr, "%s: expected %zu bytes, read %zu\n", path.c_str(), bytes, got);
  return got == bytes;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <directory>\n", argv[0]);
    return 2;
  }
  const std::string dir = std::string(argv[1]) + "/";
  int bits = 0, group = 0;
  {
    FILE* f = fopen((dir + "shape.bin").c_str(), "r");
    char c;
    if (!f || fscanf(f, "%d%c%d%c%d%c%d", &MATRIX_K, &c, &MATRIX_N, &c, &bits, &c, &group) != 7) {
      fprintf(stderr, "shape.bin must hold K,N,bits,group\n");
      return 2;
    }
    fclose(f);
  }
  MATRIX_M = 1;
  printf("M: %d, K: %d, N: %d\n", MATRIX_M, MATRIX_K, MATRIX_N);
  const size_t qsize_n = ((size_t)MATRIX_N * bits + 31) / 32, qsize_k = ((size_t)MATRIX_K * bits + 31) / 32;
  const size_t group_k = ((size_t)MATRIX_K + group - 1) / group;
  std::vector<char> input, qweight, scale, qzero, f16weight, out;
  SampleData data = {};
  data.bits = bits;
  data.group_size = group;
  data.size.input_size = (int)((size_t)MATRIX_M * MATRIX_K * sizeof(short));
  data.size.qweight_size = (int)((size_t)MATRIX_N * qsize_k * sizeof(int32_t));
  data.size.scale_size = (int)((size_t)MATRIX_N * group_k * sizeof(short));
  data.size.qzero_size = (int)(qsize_n * group_k * sizeof(int32_t));
  data.size.f16weight_size = (int)((size_t)MATRIX_N * MATRIX_K * sizeof(short));
  data.size.out_size = (int)((size_t)MATRIX_M * MATRIX_N * sizeof(short));
  if (!read_file(dir + "input.bin", &input, data.size.input_size) || !read_file(dir + "qweight.bin", &qweight, data.size.qweight_size) ||
      !read_file(dir + "scales.bin", &scale, data.size.scale_size) || !read_file(dir + "qzeros.bin", &qzero, data.size.qzero_size) ||
      !read_file(dir + "weight.bin", &f16weight, data.size.f16weight_size) || !read_file(dir + "out.bin", &out, data.size.out_size))
    return 2;
  data.input = reinterpret_cast<short*>(input.data());
  data.qweight = reinterpret_cast<int32_t*>(qweight.data());
  data.scale = reinterpret_cast<short*>(scale.data());
  data.qzero = reinterpret_cast<int32_t*>(qzero.data());
  data.f16weight = reinterpret_cast<short*>(f16weight.data());
  data.out = reinterpret_cast<short*>(out.data());
  return QbitGemv(&data);
}
