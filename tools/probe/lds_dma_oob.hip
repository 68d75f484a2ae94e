// Probe: does an out-of-range lane of `buffer_load_dword ... lds` write 0.0 to LDS on gfx950?
// (conv staging relies on it for zero padding).  Build: hipcc --offload-arch=gfx950 -O2 lds_dma_oob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, int n, int soff) {
  __shared__ float smem[512];
  smem[threadIdx.x] = -7.f; smem[threadIdx.x + 256] = -7.f;
  __syncthreads();
  auto r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned voff = (lane % 3 == 2) ? 0x80000000u : (unsigned)(threadIdx.x * 4);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 64), 4, voff, soff, 0, 0);
  // 16-byte flavour: lane writes 4 floats
  if (wave == 0) {
    unsigned v4 = (lane % 5 == 4) ? 0x80000000u : (unsigned)(lane * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 256), 16, v4, soff, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  y[threadIdx.x] = smem[threadIdx.x];
  y[threadIdx.x + 256] = smem[threadIdx.x + 256];
}
int main() {
  int n = 300;   // lanes >= 300 are naturally out of range too
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  float *x, *y;
  hipMalloc(&x, 4096); hipMalloc(&y, 2048);
  hipMemcpy(x, h.data(), 4096, hipMemcpyHostToDevice);
  for (int soff : {0, 64}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, y, n, soff);
    std::vector<float> o(512);
    hipMemcpy(o.data(), y, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
      int lane = t & 63;
      long idx = t + soff / 4;
      float want = (lane % 3 == 2 || idx >= n) ? 0.f : (float)(idx + 1);
      if (o[t] != want) { if (bad < 8) printf("soff %d dword t=%d got %g want %g\n", soff, t, o[t], want); ++bad; }
    }
    for (int t = 0; t < 256; ++t) {
      int lane = t / 4;
      long idx = t + soff / 4;
      float want = (lane % 5 == 4 || idx >= n) ? 0.f : (float)(idx + 1);
      // a 16-byte access that straddles the end is out of range as a whole or in part; report only
      if (o[256 + t] != want) { if (bad < 16) printf("soff %d x4 t=%d got %g want %g\n", soff, t, o[256 + t], want); ++bad; }
    }
    printf("soff %d: %d mismatches\n", soff, bad);
  }
  return 0;
}
