// Hardware probe: prints what ds_read_b64_tr_b16 returns per lane when lds[i] = i (fp16 index
// pattern as raw u16) and lane l supplies byte address 8*l.  Used to pin the LDS transpose-read
// semantics on gfx950 before relying on it in the attention kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  u16x4 t;
  unsigned addr = (unsigned)(size_t)lds + threadIdx.x * 8;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = t[j];
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  return 0;
}
