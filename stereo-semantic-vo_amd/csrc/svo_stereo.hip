// svo_stereo.hip - sparse epipolar stereo association + depth for gfx950.
//
// Stands where frame::MB + frame::computekeypoint_r + frame::disp2Depth stand in
// the reference (src/Tracking.cc:226-228; src/frame.cc:82-91,122-164): it turns a
// stereo pair into per-keypoint right-image x and depth.  The algorithm is the
// sparse matcher BASELINE.json's north_star names ("ComputeStereoMatches"):
// row-band candidates, Hamming argmin, 11x11 SAD refinement +-5 px at the
// keypoint's level, parabola sub-pixel, median-based outlier cut.
// The dense helpers disp2Depth / UnprojectStereo are kept as kernels too.
#include "svo_internal.h"
#include "svo_wave.h"

#define TH_HIGH 100
#define TH_ORB ((100 + 50) / 2)
#define SAD_W 5
#define SAD_L 5
#define KP_PER_WG 64
#define MAXKP_LDS 512
#define ST_NB 512   // row buckets of the right image's keypoints (below)

struct StereoSrc {
  const uint8_t* L;
  const uint8_t* R;
  int stride;
  int B;
  const uint8_t* pyr;
};

__device__ __forceinline__ const uint8_t* st_level_ptr(const SvoGeom& g, const StereoSrc& s,
                                                        int img, int l, int* pitch) {
  if (l == 0) {
    *pitch = s.stride;
    return img < s.B ? s.L + (size_t)img * s.stride * g.H
                     : s.R + (size_t)(img - s.B) * s.stride * g.H;
  }
  *pitch = g.pitch[l];
  return s.pyr + (size_t)img * g.pyr_bytes + g.loff[l];
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return wave_min_u32_dpp(v); }

// One workgroup = KP_PER_WG left keypoints of one pair (4 waves x KP_PER_WG/4 keypoints); the right image's keypoint
// bands and descriptors are staged in LDS once per workgroup.  A wave takes its keypoints in groups of ST_G: the
// group's keypoints and descriptors are requested together, then the Hamming search runs for all of them, then the
// SAD windows of all of them are requested together - two memory round trips per group instead of two per keypoint.
#define ST_G 4
__global__ __launch_bounds__(256) void k_stereo_match(SvoGeom g, StereoSrc s, const svo_kp* kp,
                                                      const uint8_t* desc, const int32_t* nkp,
                                                      int max_kp, float bf, float fx, float* uR,
                                                      float* depth, int32_t* sad_out) {
  __shared__ uint32_t rdesc[MAXKP_LDS * 8];
  __shared__ int16_t rminr[MAXKP_LDS], rmaxr[MAXKP_LDS];
  __shared__ int8_t roct[MAXKP_LDS];
  __shared__ float rx[MAXKP_LDS];
  __shared__ int sadbuf[4][128];
  // The right keypoints bucketed by row (counting sort, once per workgroup): a left keypoint's candidates lie within
  // 2 * scale[octave] + 1 rows of its own, i.e. in a few buckets - one pass of the wave over ~30 candidates instead of eight
  // passes over all 500 with the band test failing for 97 % of them.  The argmin packs the ORIGINAL index, so the winner (ties:
  // lowest index) is the one the full scan finds.
  __shared__ int bcur[ST_NB + 1];      // bucket b: order[bstart[b] .. bstart[b + 1]) (bcur: the fill cursors, then unused)
  __shared__ int bstart[ST_NB + 1];
  __shared__ uint16_t order[MAXKP_LDS];
  __shared__ uint64_t winL[4][ST_G][11][2];   // per wave and keypoint: 11 rows x 16 bytes of the left SAD window
  __shared__ uint64_t winR[4][ST_G][11][3];   // per wave and keypoint: 11 rows x 24 bytes of the right search band
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // XCD-aware order (as k_fast / k_describe): the workgroups of one pair - same right-image bands, SAD windows in the same
  // rows - run on ONE XCD and share its L2
  int pair, chunk;
  {
    const uint32_t total_wg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t xcd = lin & 7u, idx = lin >> 3, q = total_wg >> 3, r = total_wg & 7u;
    const uint32_t mapped = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    pair = (int)(mapped / gridDim.x);
    chunk = (int)(mapped - (uint32_t)pair * gridDim.x);
  }
  const int imgL = pair, imgR = s.B + pair;
  const int nL = nkp[imgL], nR = min(nkp[imgR], MAXKP_LDS);
  const svo_kp* kpL = kp + (size_t)imgL * max_kp;
  const svo_kp* kpR = kp + (size_t)imgR * max_kp;
  const uint32_t* dL = reinterpret_cast<const uint32_t*>(desc + (size_t)imgL * max_kp * 32);
  const uint32_t* dR = reinterpret_cast<const uint32_t*>(desc + (size_t)imgR * max_kp * 32);
  for (int i = tid; i < nR * 8; i += 256) rdesc[i] = dR[i];
  for (int i = tid; i < nR; i += 256) {
    const svo_kp k = kpR[i];
    const float r = 2.0f * g.scale[k.octave];
    rmaxr[i] = (int16_t)(int)ceilf(k.y + r);
    rminr[i] = (int16_t)(int)floorf(k.y - r);
    roct[i] = (int8_t)k.octave;
    rx[i] = k.x;
  }
  int bshift = 0;
  while ((g.H >> bshift) >= ST_NB) ++bshift;
  const int band = (int)ceilf(2.0f * g.scale[SVO_NLEVELS - 1]) + 1;   // |row - (int)y| of a candidate never exceeds this
  for (int i = tid; i <= ST_NB; i += 256) bcur[i] = 0;
  __syncthreads();
  auto bucket_of = [&](int i) { return min(max((int)kpR[i].y, 0), g.H - 1) >> bshift; };
  for (int i = tid; i < nR; i += 256) atomicAdd(&bcur[bucket_of(i) + 1], 1);
  __syncthreads();
  if (tid < 64) {   // exclusive prefix over the ST_NB + 1 counters: 8 (+1) per lane, then across the wave
    int loc[ST_NB / 64], sum = 0;
#pragma unroll
    for (int k = 0; k < ST_NB / 64; ++k) { loc[k] = bcur[1 + tid * (ST_NB / 64) + k]; sum += loc[k]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (tid >= o) incl += v; }
    int run = incl - sum;
    if (tid == 0) bstart[0] = 0;
#pragma unroll
    for (int k = 0; k < ST_NB / 64; ++k) { run += loc[k]; bstart[1 + tid * (ST_NB / 64) + k] = run; }
  }
  __syncthreads();
  for (int i = tid; i <= ST_NB; i += 256) bcur[i] = bstart[i];
  __syncthreads();
  for (int i = tid; i < nR; i += 256) order[atomicAdd(&bcur[bucket_of(i)], 1)] = (uint16_t)i;
  __syncthreads();
  const float maxD = fx;
  typedef uint64_t __attribute__((aligned(1))) u64u;
  for (int q0 = 0; q0 < KP_PER_WG / 4; q0 += ST_G) {
    const int iL0 = chunk * KP_PER_WG + wv * (KP_PER_WG / 4) + q0;
    if (iL0 >= max_kp) break;
    // (1) the group's keypoints and descriptors (indices clamped: every load is unconditional)
    svo_kp kl[ST_G];
    uint32_t ql[ST_G][8];
#pragma unroll
    for (int j = 0; j < ST_G; ++j) {
      const int ic = min(iL0 + j, max_kp - 1);
      kl[j] = kpL[ic];
#pragma unroll
      for (int k = 0; k < 8; ++k) ql[j][k] = dL[(size_t)ic * 8 + k];
    }
    // (2) Hamming search in the row band
    uint32_t best[ST_G];
#pragma unroll
    for (int j = 0; j < ST_G; ++j) {
      const int levelL = kl[j].octave;
      const float uL = kl[j].x;
      const int row = (int)kl[j].y;
      const float minU = uL - maxD, maxU = uL;
      uint32_t b = ((uint32_t)TH_HIGH << 16) | 0xffffu;
      if (iL0 + j < nL && maxU >= 0) {
        const int rc = min(max(row, 0), g.H - 1);
        const int c0 = bstart[max(rc - band, 0) >> bshift], c1 = bstart[(min(rc + band, g.H - 1) >> bshift) + 1];
        for (int c = c0 + lane; c < c1; c += 64) {
          const int iR = order[c];
          const int oc = roct[iR];
          const float u = rx[iR];
          const bool ok = row >= rminr[iR] && row <= rmaxr[iR] && oc >= levelL - 1 &&
                          oc <= levelL + 1 && u >= minU && u <= maxU;
          if (ok) {
            int dist = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) dist += __popc(ql[j][k] ^ rdesc[iR * 8 + k]);
            b = min(b, ((uint32_t)dist << 16) | (uint32_t)iR);
          }
        }
      }
      best[j] = wave_min_u32(b);
    }
    // (3) SAD windows of the whole group: lanes 0..10 fetch the left rows (11 px at su-5), lanes 16..26 the right
    // rows (21 px at sr0-10); coordinates are clamped into the level so that the loads need no condition
    bool refine[ST_G];
    int sr0s[ST_G];
    uint64_t wreg[ST_G][3];
#pragma unroll
    for (int j = 0; j < ST_G; ++j) {
      const int bestDist = (int)(best[j] >> 16), bestIdxR = (int)(best[j] & 0xffffu);
      const int levelL = min(max(kl[j].octave, 0), SVO_NLEVELS - 1);
      const float inv = 1.0f / g.scale[levelL];
      const float uR0 = rx[min(bestIdxR, MAXKP_LDS - 1)];
      const int su = (int)roundf(kl[j].x * inv), sv = (int)roundf(kl[j].y * inv), sr0 = (int)roundf(uR0 * inv);
      const int lw = g.w[levelL], lh = g.h[levelL];
      const bool inb = !(sv - SAD_W < 0 || sv + SAD_W >= lh || su - SAD_W < 0 || su + SAD_W >= lw ||
                         sr0 - SAD_L - SAD_W < 0 || sr0 + SAD_L + SAD_W >= lw);
      refine[j] = iL0 + j < nL && bestDist < TH_ORB && bestIdxR != 0xffff && inb;
      sr0s[j] = sr0;
      const int svc = refine[j] ? sv : SAD_W, suc = refine[j] ? su : SAD_W, src = refine[j] ? sr0 : SAD_L + SAD_W;
      int pl, pr;
      const uint8_t* IL = st_level_ptr(g, s, imgL, levelL, &pl);
      const uint8_t* IR = st_level_ptr(g, s, imgR, levelL, &pr);
      const int rowi = lane < 16 ? min(lane, 10) : min(lane - 16, 10);
      const uint8_t* p = lane < 16 ? IL + (size_t)(svc + rowi - SAD_W) * pl + suc - SAD_W
                                   : IR + (size_t)(svc + rowi - SAD_W) * pr + src - SAD_L - SAD_W;
      wreg[j][0] = *reinterpret_cast<const u64u*>(p);
      wreg[j][1] = *reinterpret_cast<const u64u*>(p + 8);
      wreg[j][2] = *reinterpret_cast<const u64u*>(p + (lane < 16 ? 8 : 16));
    }
#pragma unroll
    for (int j = 0; j < ST_G; ++j) {
      if (lane < 11) {
        winL[wv][j][lane][0] = wreg[j][0]; winL[wv][j][lane][1] = wreg[j][1];
      } else if (lane >= 16 && lane < 27) {
        winR[wv][j][lane - 16][0] = wreg[j][0]; winR[wv][j][lane - 16][1] = wreg[j][1]; winR[wv][j][lane - 16][2] = wreg[j][2];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // (4) SAD refinement, parabola, outputs
#pragma unroll
    for (int j = 0; j < ST_G; ++j) {
      const int iL = iL0 + j;
      if (iL >= max_kp) break;
      const size_t o = (size_t)pair * max_kp + iL;
      float out_u = -1.f, out_d = -1.f;
      int out_sad = -1;
      if (refine[j]) {
        const int levelL = kl[j].octave;
        const float uL = kl[j].x;
        const float sc = g.scale[levelL];
        const int sr0 = sr0s[j];
        const uint8_t* wl = reinterpret_cast<const uint8_t*>(&winL[wv][j][0][0]);   // row pitch 16
        const uint8_t* wr = reinterpret_cast<const uint8_t*>(&winR[wv][j][0][0]);   // row pitch 24
        const int cL = wl[5 * 16 + 5];
        // (inc, dy) pairs: 121 partial row sums
        for (int p = lane; p < 121; p += 64) {
          const int inc = p / 11 - SAD_L, dy = p % 11;
          const int cR = wr[5 * 24 + 10 + inc];
          const uint8_t* a = wl + dy * 16;
          const uint8_t* b = wr + dy * 24 + 5 + inc;
          int sum = 0;
#pragma unroll
          for (int dx = 0; dx < 11; ++dx) sum += abs(((int)a[dx] - cL) - ((int)b[dx] - cR));
          sadbuf[wv][p] = sum;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        int mysad = 0x7fffffff;
        if (lane < 11) {
          mysad = 0;
#pragma unroll
          for (int k = 0; k < 11; ++k) mysad += sadbuf[wv][lane * 11 + k];
        }
        int dists[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) dists[k] = __builtin_amdgcn_readlane(mysad, k);   // lanes 0..10 hold the 11 SADs
        int bsad = 0x7fffffff, binc = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k)
          if (dists[k] < bsad) { bsad = dists[k]; binc = k - SAD_L; }
        if (binc != -SAD_L && binc != SAD_L) {
          float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
          for (int k = 1; k < 10; ++k)
            if (k - SAD_L == binc) { d1 = (float)dists[k - 1]; d2 = (float)dists[k]; d3 = (float)dists[k + 1]; }
          const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
          if (!(deltaR < -1 || deltaR > 1)) {
            float bestuR = sc * ((float)sr0 + (float)binc + deltaR);
            float disparity = uL - bestuR;
            if (disparity >= 0.f && disparity < maxD) {
              if (disparity <= 0) { disparity = 0.01f; bestuR = uL - 0.01f; }
              out_d = bf / disparity;
              out_u = bestuR;
              out_sad = bsad;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) { uR[o] = out_u; depth[o] = out_d; sad_out[o] = out_sad; }
    }
    __builtin_amdgcn_wave_barrier();   // the windows are rewritten by the next group
    __threadfence_block();
  }
}

// Median-based outlier cut over one pair: median = the (nd/2)-th smallest best SAD
// among accepted keypoints; reject sad >= 1.5f*1.4f*median.
__global__ __launch_bounds__(256) void k_stereo_median(int max_kp, float* uR, float* depth,
                                                       const int32_t* sad) {
  __shared__ int ssad[1024];
  __shared__ __attribute__((aligned(16))) int skey[1024];
  __shared__ int nd, smed;
  const int tid = threadIdx.x, pair = blockIdx.x;
  const size_t base = (size_t)pair * max_kp;
  if (tid == 0) { nd = 0; smed = -1; }
  __syncthreads();
  int cnt = 0;
  for (int i = tid; i < max_kp; i += 256) {
    const int v = sad[base + i];
    ssad[i] = v;
    cnt += v >= 0;
  }
  if (cnt) atomicAdd(&nd, cnt);
  __syncthreads();
  const int n = nd;
  if (n == 0) return;
  // rank of each accepted SAD among the accepted ones (ties by index).  Rejected entries (-1) are mapped to INT_MAX so
  // that one unsigned-free comparison serves; the array is read four entries per LDS instruction, several in flight.
  for (int i = tid; i < 1024; i += 256) {
    const int v = i < max_kp ? ssad[i] : -1;
    skey[i] = v < 0 ? 0x7fffffff : v;
  }
  __syncthreads();
  const int4* skey4 = reinterpret_cast<const int4*>(skey);
  const int n4 = (max_kp + 3) / 4;
  for (int i = tid; i < max_kp; i += 256) {
    const int v = ssad[i];
    if (v < 0) continue;
    int rank = 0;
#pragma unroll 4
    for (int j4 = 0; j4 < n4; ++j4) {
      const int4 u = skey4[j4];
      const int j = 4 * j4;
      rank += (u.x < v || (u.x == v && j < i)) + (u.y < v || (u.y == v && j + 1 < i)) + (u.z < v || (u.z == v && j + 2 < i)) +
              (u.w < v || (u.w == v && j + 3 < i));
    }
    if (rank == n / 2) smed = v;
  }
  __syncthreads();
  const float thDist = 2.1f * (float)smed;
  for (int i = tid; i < max_kp; i += 256) {
    const int v = ssad[i];
    if (v >= 0 && !((float)v < thDist)) { uR[base + i] = -1.f; depth[base + i] = -1.f; }
  }
}

// frame::disp2Depth (reference src/frame.cc:140-164), dense float map.
__global__ void k_disp2depth(const float* __restrict__ disp, int count, float bf,
                             float* __restrict__ depth) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float d = disp[i];
    depth[i] = d != 0.f ? bf / d : -1.f;
  }
}

// frame::UnprojectStereo (reference src/frame.cc:166-180).
__global__ void k_unproject(const float* __restrict__ uvz, int n, svo_camera cam, const float* Rwc,
                            const float* twc, float* __restrict__ xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float u = uvz[3 * i], v = uvz[3 * i + 1], z = uvz[3 * i + 2];
  if (z > 0) {
    const float x = (u - cam.cx) * z * (1 / cam.fx);
    const float y = (v - cam.cy) * z * (1 / cam.fy);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double acc = (double)Rwc[3 * r] * (double)x + (double)Rwc[3 * r + 1] * (double)y +
                         (double)Rwc[3 * r + 2] * (double)z;
      xyz[3 * i + r] = (float)(acc + (double)twc[r]);
    }
  } else {
    xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = __builtin_nanf("");
  }
}

int svo_launch_stereo(svo_ctx* ctx, const uint8_t* d_grayL, const uint8_t* d_grayR, int stride,
                      int B, const svo_camera* cam) {
  if (B > ctx->max_batch || ctx->max_kp > 1024) return SVO_E_CAPACITY;
  StereoSrc s{d_grayL, d_grayR, stride, B, ctx->d_pyr};
  {
    SvoTimer t(ctx, "k_stereo_match");
    hipLaunchKernelGGL(k_stereo_match, dim3((ctx->max_kp + KP_PER_WG - 1) / KP_PER_WG, B), dim3(256),
                       0, ctx->stream, ctx->g, s, ctx->d_kp, ctx->d_desc, ctx->d_nkp, ctx->max_kp,
                       cam->bf, cam->fx, ctx->d_uR, ctx->d_depth, ctx->d_sad);
  }
  {
    SvoTimer t(ctx, "k_stereo_median");
    hipLaunchKernelGGL(k_stereo_median, dim3(B), dim3(256), 0, ctx->stream, ctx->max_kp, ctx->d_uR,
                       ctx->d_depth, ctx->d_sad);
  }
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_launch_disp2depth(svo_ctx* ctx, const float* disp, int count, float bf, float* depth) {
  const int blocks = min((count + 255) / 256, 2048);
  hipLaunchKernelGGL(k_disp2depth, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, ctx->stream, disp,
                     count, bf, depth);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_launch_unproject(svo_ctx* ctx, const float* uvz, int n, const svo_camera* cam,
                         const float* Rwc, const float* twc, float* xyz) {
  if (n <= 0) return SVO_OK;
  hipLaunchKernelGGL(k_unproject, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, uvz, n, *cam,
                     Rwc, twc, xyz);
  SVO_HIP(ctx, hipGetLastError());
  return SVO_OK;
}
