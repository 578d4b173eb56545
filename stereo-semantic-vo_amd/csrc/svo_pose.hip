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

__global__ __launch_bounds__(256) void k_pnp_ransac(const double* __restrict__ Xw,
                                                    const double* __restrict__ obs, int n,
                                                    const double* __restrict__ Kp,
                                                    const double* __restrict__ Tprior, uint64_t seed,
                                                    double* T, uint8_t* inlier_mask,
                                                    svo_pnp_stats* stats, const int* n_ptr,
                                                    const int* skip_ptr, const int* frame_ptr,
                                                    int use_mfma, size_t seq_stride) {
  if (blockIdx.y) {
    const size_t off = (size_t)blockIdx.y * seq_stride;
    Xw = svo_byte_offset(Xw, off); obs = svo_byte_offset(obs, off); Kp = svo_byte_offset(Kp, off);
    Tprior = svo_byte_offset(Tprior, off); T = svo_byte_offset(T, off); stats = svo_byte_offset(stats, off);
    n_ptr = svo_byte_offset(n_ptr, off); skip_ptr = svo_byte_offset(skip_ptr, off);
    frame_ptr = svo_byte_offset(frame_ptr, off);
  }
  if (n_ptr) n = *n_ptr;
  if (frame_ptr) seed = 0x5EED0000ULL + (uint64_t)*frame_ptr;
  if (skip_ptr && *skip_ptr) {   // frame 0: no PnP, the pose stays at the prior
    if (threadIdx.x < 16) T[threadIdx.x] = Tprior[threadIdx.x];
    return;
  }
  PoseLds& L = *reinterpret_cast<PoseLds*>(pose_smem);
  pnp_ransac_block(L, Xw, obs, n, Kp, Tprior, seed, T, inlier_mask, stats, use_mfma);
}

// PoseLds is larger than the 64 KB a kernel gets by default: opt in once per context (per-device attribute)
int svo_pose_lds_optin(svo_ctx* ctx) {
  if (ctx->pose_lds_state == 0) {
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_opt), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sizeof(PoseLds)) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_pnp_ransac), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(PoseLds)) == hipSuccess;
    ctx->pose_lds_state = ok ? 1 : -1;
    if (!ok) ctx->last_error = std::string("hipFuncSetAttribute(pose kernels): ") + hipGetErrorString(hipGetLastError());
  }
  return ctx->pose_lds_state > 0 ? SVO_OK : SVO_E_HIP;
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

int svo_launch_pnp(svo_ctx* ctx, const double* Xw, const double* obs, int n, const double* K,
                   const double* Tprior, uint64_t seed, double* T, uint8_t* mask,
                   svo_pnp_stats* stats) {
  if (n > PNP_MAXN) return SVO_E_CAPACITY;
  int rc = svo_pose_lds_optin(ctx);
  if (rc) return rc;
  SvoTimer tm(ctx, "k_pnp_ransac");
  hipLaunchKernelGGL(k_pnp_ransac, dim3(1), dim3(256), sizeof(PoseLds), ctx->stream, Xw, obs, n, K, Tprior, seed,
                     T, mask, stats, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr,
                     ctx->opt_pose_mfma, (size_t)0);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

