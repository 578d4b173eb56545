// svo_pose.hip - stand-alone kernels + launchers of the pose stages (device code in svo_pose_dev.h).
//
//   k_pose_opt   <- Optimizer::PoseOptimization (reference src/Optimizer.cc:15-86)
//   k_pnp_ransac <- the cv::solvePnPRansac call of pnpmatch::poseEstimationPnP (src/pnpmatch.cc:212-247)
// One 256-thread workgroup per problem; blockIdx.y selects the sequence of a multi-sequence caller.
#include "svo_pose_dev.h"

extern __shared__ __attribute__((aligned(16))) unsigned char pose_smem[];

__global__ __launch_bounds__(256) void k_pose_opt(const double* __restrict__ Xw,
                                                  const double* __restrict__ obs, int n,
                                                  const double* __restrict__ Kp, double* T,
                                                  svo_lm_stats* stats, const int* n_ptr,
                                                  int round_in_f32, int use_mfma, size_t seq_stride) {
  if (blockIdx.y) {   // sequence blockIdx.y of a multi-sequence caller
    const size_t off = (size_t)blockIdx.y * seq_stride;
    Xw = svo_byte_offset(Xw, off); obs = svo_byte_offset(obs, off); Kp = svo_byte_offset(Kp, off);
    T = svo_byte_offset(T, off); stats = svo_byte_offset(stats, off); n_ptr = svo_byte_offset(n_ptr, off);
  }
  if (n_ptr) n = *n_ptr;
  PoseLds& L = *reinterpret_cast<PoseLds*>(pose_smem);
  pose_opt_block(L, Xw, obs, n, Kp, T, stats, round_in_f32, use_mfma);
}

// ---- cv::solvePnPRansac on plain arrays (svo_pnp_ransac) --------------------------------------------------------
// k_pnp_hyp: the 100 RANSAC samples, one EPnP per wave, each with its consensus.  ONE wave per workgroup, i.e. one per
// compute unit: the float64 pipeline is shared by the four SIMDs of a CU (measured: the scalar float64 stage of the solve
// takes 18 k cycles with one wave on the CU, 50 k with four), and with 256 CUs there is no reason to share it.
// k_pnp_select: the sequential acceptance rule over those samples, the winner's pose, inlier mask and stats.
struct PnpHypLds {
  double Xw[PNP_MAXN * 3], uv[PNP_MAXN * 2];
  EpnpWaveLds ws[4];
};

__global__ __launch_bounds__(256) void k_pnp_hyp(const double* __restrict__ Xw, const double* __restrict__ obs, int n,
                                                 const double* __restrict__ Kp, const uint16_t* __restrict__ subset,
                                                 PnpHyp* hyp) {
  PnpHypLds& L = *reinterpret_cast<PnpHypLds*>(pose_smem);
  for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) L.Xw[i] = Xw[i];
  for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) L.uv[i] = obs[i];
  __syncthreads();
  const double K[4] = {Kp[0], Kp[1], Kp[2], Kp[3]};
  pnp_hyp_block(L.ws, L.Xw, L.uv, n, K, subset, hyp, blockIdx.x * (blockDim.x >> 6));
}

// parity mode (svo_set_option "epnp_exact"): one single-wave workgroup per sample, OpenCV's loops in order on lane 0
__global__ __launch_bounds__(64) void k_pnp_hyp_exact(const double* __restrict__ Xw, const double* __restrict__ obs, int n,
                                                      const double* __restrict__ Kp, const uint16_t* __restrict__ subset,
                                                      PnpHyp* hyp) {
  __shared__ PnpExactLds S;
  const double K[4] = {Kp[0], Kp[1], Kp[2], Kp[3]};
  pnp_hyp_exact_wave(S, Xw, obs, n, K, subset, hyp, blockIdx.x);
}

// order-preserving wave mode ("epnp_exact" = 2): one single-wave workgroup per sample, OpenCV's operations over the wavefront
__global__ __launch_bounds__(64) void k_pnp_hyp_ord(const double* __restrict__ Xw, const double* __restrict__ obs, int n,
                                                    const double* __restrict__ Kp, const uint16_t* __restrict__ subset,
                                                    PnpHyp* hyp, int force_seq) {
  __shared__ PnpOrdLds S;
  const double K[4] = {Kp[0], Kp[1], Kp[2], Kp[3]};
  pnp_hyp_ord_wave(S, Xw, obs, n, K, subset, hyp, blockIdx.x, force_seq != 0);
}

__global__ __launch_bounds__(256) void k_pnp_select(const double* __restrict__ Xw, const double* __restrict__ obs, int n,
                                                    const double* __restrict__ Kp, const double* __restrict__ Tfallback,
                                                    const PnpHyp* hyp, double* T, uint8_t* inlier_mask,
                                                    svo_pnp_stats* stats) {
  __shared__ int cnt[PNP_HYP], ok[PNP_HYP], s_best, s_good, s_iters;
  const int tid = threadIdx.x;
  if (tid < PNP_HYP) { cnt[tid] = n >= 5 ? hyp[tid].cnt : 0; ok[tid] = n >= 5 ? hyp[tid].ok : 0; }
  __syncthreads();
  if (tid == 0) {
    int good = 0, iters = 0;
    s_best = n >= 5 ? pnp_select(cnt, ok, n, &good, &iters) : -1;
    s_good = good; s_iters = iters;
  }
  __syncthreads();
  const int best = s_best;
  if (best >= 0) {
    const PnpHyp h = hyp[best];
    const double K[4] = {Kp[0], Kp[1], Kp[2], Kp[3]};
    if (inlier_mask)
      for (int e = tid; e < n; e += 256)
        inlier_mask[e] = n == 5 ? 1 : (pnp_inlier(h.R, h.t, Xw + 3 * e, obs + 2 * e, K) ? 1 : 0);
    if (tid < 16) {
      const int r = tid >> 2, c = tid & 3;
      T[tid] = r == 3 ? (c == 3 ? 1.0 : 0.0) : (c == 3 ? h.t[r] : h.R[3 * r + c]);
    }
  } else {
    if (inlier_mask)
      for (int e = tid; e < n; e += 256) inlier_mask[e] = 0;
    if (tid < 16) T[tid] = Tfallback[tid];
  }
  if (tid == 0 && stats) {
    stats->n_points = n; stats->n_inliers = best >= 0 ? s_good : 0; stats->best_hypothesis = best;
    stats->ok = best >= 0 ? 1 : 0; stats->iterations = s_iters;
  }
}

// parity probe: one EPnP on five correspondences, one wave (timing experiments: `waves` waves solve the same sample side
// by side, `reps` times each; wave 0 reports)
__global__ __launch_bounds__(256) void k_epnp5_probe(const double* X5, const double* u5, const double* Kp, double* Rt, int reps) {
  __shared__ EpnpWaveLds wsv[4];
  EpnpWaveLds& ws = wsv[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  double K[4] = {Kp[0], Kp[1], Kp[2], Kp[3]}, R[9], t[3];
  if (lane < 15) ws.x5[lane] = X5[lane];
  if (lane < 10) ws.u5[lane] = u5[lane];
  bool ok = false;
  for (int rep = 0; rep < reps; ++rep) ok = epnp5_wave(ws, K, R, t);   // reps > 1: timing with a warm instruction cache
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    for (int i = 0; i < 9; ++i) Rt[i] = R[i];
    Rt[9] = t[0]; Rt[10] = t[1]; Rt[11] = t[2]; Rt[12] = ok ? 1.0 : 0.0;
    for (int b = 0; b < 3; ++b) Rt[13 + b] = ws.out[b][12];
    for (int b = 0; b < 4; ++b) Rt[16 + b] = (double)(ws.stamp[b + 1] - ws.stamp[b]);
    Rt[20] = ws.sweeps;
    Rt[21] = (double)(ws.stamp[5] - ws.stamp[3]); Rt[22] = (double)(ws.stamp[6] - ws.stamp[5]); Rt[23] = (double)(ws.stamp[7] - ws.stamp[6]);
  }
}
__global__ __launch_bounds__(64) void k_epnp5_probe_exact(const double* X5, const double* u5, const double* Kp, double* Rt) {
  __shared__ epnp_exact::Work W;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double K[4] = {Kp[0], Kp[1], Kp[2], Kp[3]};
  const bool ok = epnp_exact::solve5(W, X5, u5, K, Rt, Rt + 9, Rt + 13);
  Rt[12] = ok ? 1.0 : 0.0;
  for (int b = 16; b < 24; ++b) Rt[b] = 0.0;
}
// the same in the order-preserving wave mode; Rt[16..22]: ticks of the stages, Rt[23]: 1 if the sequential fallback ran
__global__ __launch_bounds__(64) void k_epnp5_probe_ord(const double* X5, const double* u5, const double* Kp, double* Rt, int force_seq) {
  __shared__ PnpOrdLds L;
  const int lane = threadIdx.x;
  if (lane < 15) L.S.pws[lane] = X5[lane];
  if (lane < 10) L.S.us[lane] = u5[lane];
  double K[4] = {Kp[0], Kp[1], Kp[2], Kp[3]}, R[9], t[3], rep[3];
  const bool ok = epnp_ord::solve5_wave(L.S, L.W, K, R, t, rep, force_seq != 0);
  if (lane == 0 && blockIdx.x == 0) {
    for (int i = 0; i < 9; ++i) Rt[i] = R[i];
    Rt[9] = t[0]; Rt[10] = t[1]; Rt[11] = t[2]; Rt[12] = ok ? 1.0 : 0.0;
    for (int b = 0; b < 3; ++b) Rt[13 + b] = rep[b];
    for (int b = 0; b < 7; ++b) Rt[16 + b] = (double)(L.S.stamp[b + 1] - L.S.stamp[b]);
    Rt[23] = (double)L.S.flag;
  }
}
int svo_launch_epnp5_probe(svo_ctx* ctx, const double* X5, const double* u5, const double* K, double* Rt, int reps) {
  if (ctx->opt_epnp_exact == 2) {
    hipLaunchKernelGGL(k_epnp5_probe_ord, dim3(1), dim3(64), 0, ctx->stream, X5, u5, K, Rt, ctx->opt_epnp_force_seq);
    SVO_HIP(ctx, hipGetLastError());
    return SVO_OK;
  }
  if (ctx->opt_epnp_exact) {
    hipLaunchKernelGGL(k_epnp5_probe_exact, dim3(1), dim3(64), 0, ctx->stream, X5, u5, K, Rt);
    SVO_HIP(ctx, hipGetLastError());
    return SVO_OK;
  }
  const char* wv = getenv("SVO_EPNP_WAVES");
  const char* gv = getenv("SVO_EPNP_GRID");
  const int waves = wv ? std::min(std::max(atoi(wv), 1), 4) : 1, grid = gv ? std::max(atoi(gv), 1) : 1;
  hipLaunchKernelGGL(k_epnp5_probe, dim3(grid), dim3(64 * waves), 0, ctx->stream, X5, u5, K, Rt, reps < 1 ? 1 : reps);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

// PoseLds is larger than the 64 KB a kernel gets by default: opt in once per context (per-device attribute)
int svo_pose_lds_optin(svo_ctx* ctx) {
  if (ctx->pose_lds_state == 0) {
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_opt), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sizeof(PoseLds)) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_pnp_hyp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(PnpHypLds)) == hipSuccess;
    ctx->pose_lds_state = ok ? 1 : -1;
    if (!ok) ctx->last_error = std::string("hipFuncSetAttribute(pose kernels): ") + hipGetErrorString(hipGetLastError());
  }
  return ctx->pose_lds_state > 0 ? SVO_OK : SVO_E_HIP;
}

// The sample indices cv::RNG(state) hands RANSACPointSetRegistrator::getSubset for `n` points: 100 samples of 5 distinct
// indices, each index redrawn until it differs from the sample's earlier ones (OpenCV 3.2 ptsetreg.cpp).  The RNG is
// re-seeded with (uint64)-1 inside every run(), so the table depends on n only.
void svo_pnp_subsets(uint64_t state, int n, uint16_t* out /*[100 * 5]*/) {
  if (!state) state = ~0ull;
  auto next = [&]() -> unsigned {
    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  };
  for (int k = 0; k < PNP_HYP; ++k) {
    int idx[5];
    for (int i = 0; i < 5; ++i) {
      if (n <= 5) { idx[i] = i < n ? i : 0; continue; }   // count == modelPoints: the points themselves, no draw
      for (;;) {
        const int c = idx[i] = (int)(next() % (unsigned)n);
        int j = 0;
        for (; j < i; ++j)
          if (c == idx[j]) break;
        if (j == i) break;
      }
    }
    for (int i = 0; i < 5; ++i) out[5 * k + i] = (uint16_t)idx[i];
  }
}

int svo_launch_pose_opt(svo_ctx* ctx, const double* Xw, const double* obs, int n, const double* K,
                        double* T, svo_lm_stats* stats) {
  int rc = svo_pose_lds_optin(ctx);
  if (rc) return rc;
  SvoTimer tm(ctx, "k_pose_opt");
  hipLaunchKernelGGL(k_pose_opt, dim3(1), dim3(256), sizeof(PoseLds), ctx->stream, Xw, obs, n, K, T, stats,
                     (const int*)nullptr, 0, ctx->opt_pose_mfma, (size_t)0);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

// device arrays in, device results out; `subset`: 100 x 5 sample indices for this n (device), `hyp`: 100 records scratch
size_t svo_pnp_hyp_bytes() { return sizeof(PnpHyp) * PNP_HYP; }

int svo_launch_pnp(svo_ctx* ctx, const double* Xw, const double* obs, int n, const double* K,
                   const double* Tfallback, const uint16_t* subset, void* hyp_v, double* T, uint8_t* mask,
                   svo_pnp_stats* stats) {
  PnpHyp* hyp = reinterpret_cast<PnpHyp*>(hyp_v);
  if (n > PNP_MAXN) return SVO_E_CAPACITY;
  int rc = svo_pose_lds_optin(ctx);
  if (rc) return rc;
  SvoTimer tm(ctx, "k_pnp_ransac");
  if (n >= 5 && ctx->opt_epnp_exact == 2)
    hipLaunchKernelGGL(k_pnp_hyp_ord, dim3(PNP_HYP), dim3(64), 0, ctx->stream, Xw, obs, n, K, subset, hyp, ctx->opt_epnp_force_seq);
  else if (n >= 5 && ctx->opt_epnp_exact)
    hipLaunchKernelGGL(k_pnp_hyp_exact, dim3(PNP_HYP), dim3(64), 0, ctx->stream, Xw, obs, n, K, subset, hyp);
  else if (n >= 5)
    hipLaunchKernelGGL(k_pnp_hyp, dim3(PNP_HYP), dim3(64), sizeof(PnpHypLds), ctx->stream, Xw, obs, n, K, subset, hyp);
  hipLaunchKernelGGL(k_pnp_select, dim3(1), dim3(256), 0, ctx->stream, Xw, obs, n, K, Tfallback, hyp, T, mask, stats);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}
