// Probe: does an out-of-range buffer_load ... lds write zeros into LDS or leave it untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const float* g, float* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) float smem[256];
  smem[threadIdx.x] = -7.0f;  // sentinel (64 threads x 4 floats = 256 floats)
  smem[threadIdx.x + 64] = -7.0f; smem[threadIdx.x + 128] = -7.0f; smem[threadIdx.x + 192] = -7.0f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  int voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0x7ffffff0;   // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = smem[threadIdx.x * 4 + i];
}
int main() {
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = 1.0f + i;
  float *g, *o; hipMalloc(&g, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(g, o, sizeof(h));
  float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("lane0 (in range): %g %g %g %g\n", r[0], r[1], r[2], r[3]);
  printf("lane1 (OOB):      %g %g %g %g\n", r[4], r[5], r[6], r[7]);
  printf("lane2 (in range): %g %g %g %g\n", r[8], r[9], r[10], r[11]);
  printf("lane3 (OOB):      %g %g %g %g\n", r[12], r[13], r[14], r[15]);
  return 0;
}
