// Input staging of the training loop for gfx950 (main_nce.py:207-209,299-302,310;
// main_coclr.py:221-223,366-368; utils/transforms.py:57-63; model/pretrain.py:149-150).
//
// The reference receives fp32 frames (B, 3, num_seq*seq_len, H, W) in [0,1] from the loader
// (403 MB/step over PCIe at B=32), normalises them on the GPU (one full-tensor pass), then
// .view().transpose(1,2).contiguous() (a second full copy) and the model .contiguous()-copies each
// clip again.  Here the loader's frames may stay uint8 (101 MB/step over PCIe) and ONE kernel does
// ToTensor's /255, Normalize(mean, std, channel=1) and the (B,C,S,T,H,W) -> (B,S,C,T,H,W)
// re-layout: 1 (or 4) bytes read + 4 bytes written per element, 16-byte accesses.  Arithmetic is
// the reference's, operation for operation (x/255, then (x-mean)/std with IEEE division), so the
// result is bit-identical to ToTensor + Normalize on the same bytes.
#include "common.h"
#include "../../include/coclr_hip.h"

#pragma clang fp contract(off)      // bit-identical to ToTensor + Normalize: every operation rounded

namespace {

struct StageArgs {
  const void* in; float* out;
  float mean[4], std[4];
  int C, S;
  long THW;        // elements of one (b, c, s) run: seq_len*H*W, contiguous on both sides
  int from_u8;
};

__device__ __forceinline__ float norm1(float x, float mean, float std) {
  return __fdiv_rn(__fsub_rn(x, mean), std);
}

template <bool U8>
__global__ void __launch_bounds__(256)
stage_clips_kernel(const StageArgs a) {
  // blockIdx.y = (b*C + c)*S + s on the source side
  const int run = blockIdx.y;
  const int s = run % a.S, bc = run / a.S;
  const int c = bc % a.C, b = bc / a.C;
  const float mean = a.mean[c], std = a.std[c];
  float* dst = a.out + (((long)b * a.S + s) * a.C + c) * a.THW;
  if (U8) {
    const uint8_t* src = static_cast<const uint8_t*>(a.in) + (long)run * a.THW;
    if ((a.THW & 15) == 0 && (((uintptr_t)src) & 15) == 0) {
      const long n16 = a.THW >> 4;
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const uint4 q = reinterpret_cast<const uint4*>(src)[i];
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float4 o;
          o.x = norm1(__fdiv_rn((float)(w[k] & 255u), 255.f), mean, std);
          o.y = norm1(__fdiv_rn((float)((w[k] >> 8) & 255u), 255.f), mean, std);
          o.z = norm1(__fdiv_rn((float)((w[k] >> 16) & 255u), 255.f), mean, std);
          o.w = norm1(__fdiv_rn((float)(w[k] >> 24), 255.f), mean, std);
          reinterpret_cast<float4*>(dst)[i * 4 + k] = o;
        }
      }
    } else {
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.THW; i += (long)gridDim.x * 256)
        dst[i] = norm1(__fdiv_rn((float)src[i], 255.f), mean, std);
    }
  } else {
    const float* src = static_cast<const float*>(a.in) + (long)run * a.THW;
    if ((a.THW & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
      const long n4 = a.THW >> 2;
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 q = reinterpret_cast<const float4*>(src)[i];
        float4 o;
        o.x = norm1(q.x, mean, std); o.y = norm1(q.y, mean, std);
        o.z = norm1(q.z, mean, std); o.w = norm1(q.w, mean, std);
        reinterpret_cast<float4*>(dst)[i] = o;
      }
    } else {
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.THW; i += (long)gridDim.x * 256)
        dst[i] = norm1(src[i], mean, std);
    }
  }
}

}  // namespace

extern "C" int coclr_stage_clips(const void* frames, int from_u8, float* out, int B, int C, int S,
                                 int64_t THW, const float* mean, const float* std, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || C > 4 || S <= 0 || THW <= 0 || !frames || !out || !mean || !std)
    return COCLR_EINVAL;
  if ((long)B * C * S > 65535) return COCLR_EINVAL;
  StageArgs a;
  a.in = frames; a.out = out; a.C = C; a.S = S; a.THW = (long)THW; a.from_u8 = from_u8;
  for (int c = 0; c < 4; ++c) {
    a.mean[c] = c < C ? mean[c] : 0.f;     // host arrays: three floats, read at call time
    a.std[c] = c < C ? std[c] : 1.f;
    if (c < C && a.std[c] == 0.f) return COCLR_EINVAL;
  }
  const long per_thread = from_u8 ? 16 : 4;
  long gx = (THW / per_thread + 255) / 256;
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  dim3 grid((unsigned)gx, (unsigned)(B * C * S));
  if (from_u8) hipLaunchKernelGGL(stage_clips_kernel<true>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(stage_clips_kernel<false>, grid, dim3(256), 0, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}
