// lds_dma_layout.hip — where do the bytes of a 12- and 16-byte-per-lane LDS-DMA land?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ void k(const uint32_t* src, uint32_t* out, int bytes) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4096, 0x00020000);
  if (bytes == 12) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds, 12, threadIdx.x * 12, 0, 0, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds, 16, threadIdx.x * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  uint32_t h[1024], *d, *o, ho[512];
  for (int i = 0; i < 1024; ++i) h[i] = i;  // dword i holds i: source dword index
  hipMalloc(&d, 4096); hipMalloc(&o, 2048);
  hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  for (int bytes : {12, 16}) {
    k<<<1, 64>>>(d, o, bytes);
    hipMemcpy(ho, o, 2048, hipMemcpyDeviceToHost);
    printf("size %d: LDS dword -> source dword\n", bytes);
    for (int i = 0; i < 256; ++i) printf("%s%4d", (i % 16 == 0) ? "\n  " : "", ho[i] == 0xDEADBEEFu ? -1 : (int)ho[i]);
    printf("\n");
  }
  return 0;
}
