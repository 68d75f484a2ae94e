// Optimizer step of the training loop (main_nce.py:190-200,331; main_coclr.py:213,406) for gfx950.
//
// The launch scripts build ONE param group per tensor (470 groups, 235 with gradients) and call
// torch.optim.Adam.step(): hundreds of small launches per step.  Here the whole step is one
// pointer-table kernel over all tensors (HBM-bound: reads p, g, m, v and writes p, m, v once =
// 28 B per parameter), with the momentum-encoder update of the NEXT forward
// (model/pretrain.py:76-80: p_k = p_k*m + p_q*(1-m)) folded into the same pass while the fresh
// p_q is still in registers (+8 B per parameter instead of a separate 12 B pass).
//
// Arithmetic follows torch.optim.Adam (amsgrad=False, maximize=False), operation for operation:
//   g' = g + wd*p;  m = m + (g'-m)*(1-b1);  v = v*b2 + (1-b2)*g'*g';
//   p  = p - (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// with the step counters t kept on the device (one float per group, torch's fused/capturable
// state format), so a step never touches the host.
#include "common.h"
#include "../../include/coclr_hip.h"
#include <math.h>

// Every operation below is rounded on its own, as the separate ATen kernels of torch.optim.Adam and
// of the reference's momentum update are: no fused multiply-add contraction in this file (HIP's
// default is -ffp-contract=fast, and __fmul_rn / __fadd_rn are plain operators on AMD).
#pragma clang fp contract(off)

namespace {

// ---------------------------------------------------------------------------
// multi-tensor momentum update: table[3*i..] = {dst ptr, src ptr, count}
// dst = dst*m + src*(1-m), rounded as two products and one add (matches the
// reference's p_k*m + p_q*(1-m) tensor expression, no FMA contraction).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
momentum_update_kernel(const int64_t* __restrict__ table, float m, float one_minus_m) {
  const int64_t* ent = table + 3 * (long)blockIdx.x;
  float* dst = reinterpret_cast<float*>(ent[0]);
  const float* src = reinterpret_cast<const float*>(ent[1]);
  const int cnt = (int)ent[2];
  int done = 0;
  if (((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0) {       // 16-byte lanes: HBM-bound pass
    const int n4 = cnt >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 d = reinterpret_cast<float4*>(dst)[i];
      const float4 s = reinterpret_cast<const float4*>(src)[i];
      d.x = __fadd_rn(__fmul_rn(d.x, m), __fmul_rn(s.x, one_minus_m));
      d.y = __fadd_rn(__fmul_rn(d.y, m), __fmul_rn(s.y, one_minus_m));
      d.z = __fadd_rn(__fmul_rn(d.z, m), __fmul_rn(s.z, one_minus_m));
      d.w = __fadd_rn(__fmul_rn(d.w, m), __fmul_rn(s.w, one_minus_m));
      reinterpret_cast<float4*>(dst)[i] = d;
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < cnt; i += 256)
    dst[i] = __fadd_rn(__fmul_rn(dst[i], m), __fmul_rn(src[i], one_minus_m));
}

// table row: int64[8] = {p, g, exp_avg, exp_avg_sq, key-param (0: none), count, group, unused}
// hyper row: double[8] = {lr, beta1, beta2, eps, weight_decay, -, -, -} (Python floats are doubles:
//            1-beta2 = 0.001 exactly as torch forms it, not 1.f - 0.999f)
__global__ void __launch_bounds__(256)
adam_multi_kernel(const int64_t* __restrict__ table, const double* __restrict__ hyper,
                  const float* __restrict__ steps, float mom_m, float mom_1m) {
  const int64_t* ent = table + 8 * (long)blockIdx.x;
  float* p = reinterpret_cast<float*>(ent[0]);
  const float* g = reinterpret_cast<const float*>(ent[1]);
  float* m = reinterpret_cast<float*>(ent[2]);
  float* v = reinterpret_cast<float*>(ent[3]);
  float* k = reinterpret_cast<float*>(ent[4]);
  const int cnt = (int)ent[5];
  const int grp = (int)ent[6];
  const double* h = hyper + 8 * (long)grp;
  const double lr = h[0], b1d = h[1], b2d = h[2];
  const float eps = (float)h[3], wd = (float)h[4], b2 = (float)b2d;
  const double t = (double)steps[grp] + 1.0;
  const double bc1 = 1.0 - pow(b1d, t), bc2 = 1.0 - pow(b2d, t);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float w1 = (float)(1.0 - b1d), w2 = (float)(1.0 - b2d);

  auto update = [&](float pv, float gv, float& mv, float& vv) -> float {
    if (wd != 0.f) gv = __fadd_rn(gv, __fmul_rn(wd, pv));
    mv = __fadd_rn(mv, __fmul_rn(__fsub_rn(gv, mv), w1));              // lerp_(g, 1-b1)
    vv = __fadd_rn(__fmul_rn(vv, b2), __fmul_rn(__fmul_rn(w2, gv), gv));  // mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vv), bc2_sqrt), eps);
    return __fsub_rn(pv, __fmul_rn(step_size, __fdiv_rn(mv, denom)));  // addcdiv_(m, denom, -step_size)
  };

  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)k) & 15) == 0);
  int done = 0;
  if (vec) {
    const int n4 = cnt >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 pv = reinterpret_cast<float4*>(p)[i];
      const float4 gv = reinterpret_cast<const float4*>(g)[i];
      float4 mv = reinterpret_cast<float4*>(m)[i];
      float4 vv = reinterpret_cast<float4*>(v)[i];
      pv.x = update(pv.x, gv.x, mv.x, vv.x);
      pv.y = update(pv.y, gv.y, mv.y, vv.y);
      pv.z = update(pv.z, gv.z, mv.z, vv.z);
      pv.w = update(pv.w, gv.w, mv.w, vv.w);
      reinterpret_cast<float4*>(p)[i] = pv;
      reinterpret_cast<float4*>(m)[i] = mv;
      reinterpret_cast<float4*>(v)[i] = vv;
      if (k) {
        float4 kv = reinterpret_cast<float4*>(k)[i];
        kv.x = __fadd_rn(__fmul_rn(kv.x, mom_m), __fmul_rn(pv.x, mom_1m));
        kv.y = __fadd_rn(__fmul_rn(kv.y, mom_m), __fmul_rn(pv.y, mom_1m));
        kv.z = __fadd_rn(__fmul_rn(kv.z, mom_m), __fmul_rn(pv.z, mom_1m));
        kv.w = __fadd_rn(__fmul_rn(kv.w, mom_m), __fmul_rn(pv.w, mom_1m));
        reinterpret_cast<float4*>(k)[i] = kv;
      }
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < cnt; i += 256) {
    float mv = m[i], vv = v[i];
    const float pn = update(p[i], g[i], mv, vv);
    p[i] = pn; m[i] = mv; v[i] = vv;
    if (k) k[i] = __fadd_rn(__fmul_rn(k[i], mom_m), __fmul_rn(pn, mom_1m));
  }
}

// steps[groups[i]] += 1 for the groups that took part in the launch above
__global__ void adam_advance_kernel(float* steps, const int32_t* __restrict__ groups, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) steps[groups[i]] += 1.f;
}

}  // namespace

extern "C" int coclr_momentum_update(const int64_t* table, int nchunks, float m, float one_minus_m,
                                     void* stream) {
  if (nchunks <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(momentum_update_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                     table, m, one_minus_m);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_adam_step(const int64_t* table, int nchunks, const double* hyper, float* steps,
                               const int32_t* groups, int ngroups, float mom_m, float mom_1m,
                               void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (nchunks <= 0 || ngroups <= 0 || !table || !hyper || !steps || !groups) return COCLR_EINVAL;
  hipLaunchKernelGGL(adam_multi_kernel, dim3(nchunks), dim3(256), 0, stream, table, hyper, steps,
                     mom_m, mom_1m);
  COCLR_LAUNCH_CHECK();
  hipLaunchKernelGGL(adam_advance_kernel, dim3(cdiv(ngroups, 256)), dim3(256), 0, stream, steps,
                     groups, ngroups);
  COCLR_LAUNCH_CHECK();
  return 0;
}
