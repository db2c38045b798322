#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 4];
  int l = threadIdx.x;
  for (int j = 0; j < 4; ++j) lds[l * 4 + j] = in[l * 4 + j];
  __syncthreads();
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + l * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short h[256], o[256];
  for (int i = 0; i < 256; ++i) h[i] = i;
  short *di, *dout;
  hipMalloc(&di, 512); hipMalloc(&dout, 512);
  hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
  return 0;
}
