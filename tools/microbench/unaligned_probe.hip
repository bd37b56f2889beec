// unaligned_probe.hip — does a buffer / global load of 8, 12 or 16 bytes per lane return the right bytes when its address is only
// 2-byte aligned (rows of an fp16 matrix with an odd row length start there), and what does it cost?  Prints, per load width and
// byte offset, whether every lane got the bytes memcpy would have given it, and the time of a streaming read at that offset.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/unaligned_probe.hip -o tools/microbench/unaligned_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int W>  // dwords per lane
__global__ void probe(const uint8_t* base, uint32_t bytes, uint32_t byte_off, uint32_t* out) {
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, (int)bytes, 0x00020000);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t off = byte_off + t * (uint32_t)(4 * W);
  uint32_t v[4] = {0, 0, 0, 0};
  if constexpr (W == 4) {
    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  } else if constexpr (W == 2) {
    const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    v[0] = x.x; v[1] = x.y;
  } else {
    v[0] = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
  }
  for (int i = 0; i < W; ++i) out[t * W + i] = v[i];
}

template <int W>
static void run(const uint8_t* d, const std::vector<uint8_t>& h, uint32_t* dout) {
  const uint32_t threads = 64 * 256, bytes = (uint32_t)h.size();
  std::vector<uint32_t> got(threads * W), want(threads * W);
  for (uint32_t off : {0u, 2u, 4u, 6u, 1u, 10u, 126u}) {
    hipLaunchKernelGGL(probe<W>, dim3(threads / 256), dim3(256), 0, 0, d, bytes, off, dout);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
    memcpy(want.data(), h.data() + off, want.size() * 4);
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe<W>, dim3(threads / 256), dim3(256), 0, 0, d, bytes, off, dout);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("buffer_load %2d B per lane at byte offset %3u: %s (%zu of %zu dwords differ), %.2f us per launch\n", 4 * W, off, bad ? "WRONG" : "ok", bad, got.size(), ms * 5.0f);
  }
}

int main() {
  const size_t bytes = 64 * 256 * 16 + 256;
  std::vector<uint8_t> h(bytes);
  for (size_t i = 0; i < bytes; ++i) h[i] = (uint8_t)((i * 131u) ^ (i >> 8));
  uint8_t* d;
  uint32_t* dout;
  (void)hipMalloc(&d, bytes);
  (void)hipMalloc(&dout, 64 * 256 * 16);
  (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
  run<4>(d, h, dout);
  run<2>(d, h, dout);
  run<1>(d, h, dout);
  return 0;
}
