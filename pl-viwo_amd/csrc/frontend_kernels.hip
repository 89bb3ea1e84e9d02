// frontend_kernels.hip — gfx950 kernels of the point front-end (TrackKLT's OpenCV calls).
//
//   hist_kernel / equalize_kernel  K1  cv::equalizeHist            REF call site: ov_core/src/track/TrackKLT.cpp:59
//   pyrdown_kernel                 K2  cv::buildOpticalFlowPyramid REF call site: TrackKLT.cpp:71
//   lk_kernel                      K3  cv::calcOpticalFlowPyrLK    REF call site: TrackKLT.cpp:857-858
//   undistort_kernel               K6  cv::undistortPoints         REF call site: ov_core/src/cam/CamRadtan.h:99-120
//   ransac_*_kernel                K7  cv::findFundamentalMat      REF call site: TrackKLT.cpp:870-873
//
// Arithmetic contract (DESIGN.md "Front-end arithmetic"): integer image arithmetic is exact
// (LUT, 5x5 binomial, Scharr, 14-bit bilinear weights); LK's normal-equation sums are exact
// int64 wave reductions rounded to float once, so tracked positions are comparable bit-for-bit
// with the CPU oracle whatever the reduction order.
#include "radtan_core.hpp"
#include <algorithm>

#include "frontend_kernels.hpp"
#include "wave_ops.hpp"
#include "equalize_lut.hpp"

namespace plv {

__device__ __forceinline__ int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * len - 2 - p;
  if (p < 0) p = -p;  // second fold only matters for tiny levels
  return p;
}

// ------------------------------------------------------------------------------------------ K1
// COPY_IN: `img` is the library's pinned host block (the caller's image on its way in) and the kernel also leaves the image in
// `raw` (HBM) for everything behind it — the pixels cross PCIe once, as 16-byte loads of this kernel, and no copy command sits in
// front of the frame (plv_feed_image_enqueue).
template <bool COPY_IN>
__global__ void __launch_bounds__(256) hist_kernel(const uint8_t *__restrict__ img, int npix, unsigned *__restrict__ hist, uint8_t *__restrict__ raw) {
  __shared__ unsigned sh[256];
  sh[threadIdx.x] = 0;
  __syncthreads();
  const int nvec = npix >> 4;
  const uint4 *v = reinterpret_cast<const uint4 *>(img);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x) {
    uint4 q = v[i];
    if (COPY_IN) reinterpret_cast<uint4 *>(raw)[i] = q;
    unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      atomicAdd(&sh[wds[k] & 255], 1u);
      atomicAdd(&sh[(wds[k] >> 8) & 255], 1u);
      atomicAdd(&sh[(wds[k] >> 16) & 255], 1u);
      atomicAdd(&sh[wds[k] >> 24], 1u);
    }
  }
  if (blockIdx.x == 0)
    for (int i = (nvec << 4) + threadIdx.x; i < npix; i += blockDim.x) {
      const uint8_t px = img[i];
      if (COPY_IN) raw[i] = px;
      atomicAdd(&sh[px], 1u);
    }
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

// LUT (every block rebuilds it from the 256-bin histogram: 256 adds) + apply.
__global__ void __launch_bounds__(256) equalize_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int npix,
                                                       unsigned *__restrict__ hist) {
  __shared__ unsigned cdf[256];
  __shared__ uint8_t lut[256];
  __shared__ int first_bin;
  __shared__ bool last_block;
  const int t = threadIdx.x;
  cdf[t] = hist[t];
  if (t == 0) first_bin = 256;
  __syncthreads();
  if (cdf[t]) atomicMin(&first_bin, t);
  // inclusive scan (Hillis-Steele, 8 steps)
  for (int off = 1; off < 256; off <<= 1) {
    unsigned v = t >= off ? cdf[t - off] : 0;
    __syncthreads();
    cdf[t] += v;
    __syncthreads();
  }
  const int i0 = first_bin;
  const unsigned h0 = hist[i0];
  if ((int)h0 == npix) {
    lut[t] = (uint8_t)t;  // constant image: copy (cv::equalizeHist sets dst = i0 = src)
  } else {
    const float scale = (256 - 1.f) / (float)(npix - (int)h0);
    int v = 0;
    if (t > i0) v = __float2int_rn((float)(int)(cdf[t] - h0) * scale);
    lut[t] = (uint8_t)min(max(v, 0), 255);
  }
  __syncthreads();
  const int nvec = npix >> 4;
  const uint4 *s = reinterpret_cast<const uint4 *>(src);
  uint4 *d = reinterpret_cast<uint4 *>(dst);
  for (int i = blockIdx.x * blockDim.x + t; i < nvec; i += gridDim.x * blockDim.x) {
    uint4 q = s[i];
    unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      wds[k] = (unsigned)lut[wds[k] & 255] | ((unsigned)lut[(wds[k] >> 8) & 255] << 8) |
               ((unsigned)lut[(wds[k] >> 16) & 255] << 16) | ((unsigned)lut[wds[k] >> 24] << 24);
    d[i] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
  if (blockIdx.x == 0)
    for (int i = (nvec << 4) + t; i < npix; i += blockDim.x) dst[i] = lut[src[i]];
  // every workgroup has consumed the histogram before this point (LUT built behind a barrier):
  // the last one to arrive clears it and the arrival counter for the next frame, so the feed
  // path needs no memset launch (hist[256] is the counter; the buffer is zeroed once at creation)
  if (t == 0) {
    __threadfence();
    last_block = atomicAdd(&hist[256], 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last_block) {
    hist[t] = 0;
    if (t == 0) hist[256] = 0;
  }
}

// ------------------------------------------------------------------------------------------ K2
// 16x16 output pixels per workgroup; the (35 x 35) source footprint is staged in LDS once,
// filtered horizontally into an int plane, then vertically: dst = (sum + 128) >> 8.
#define PD_T 16
#define PD_S (2 * PD_T + 3)
__global__ void __launch_bounds__(256) pyrdown_kernel(const uint8_t *__restrict__ src, int sw, int sh_, uint8_t *__restrict__ dst,
                                                      int dw, int dh, unsigned *__restrict__ clear_hist) {
  __shared__ uint8_t tile[PD_S][PD_S + 1];
  __shared__ int hrow[PD_S][PD_T + 1];
  if (clear_hist && blockIdx.x == 0 && blockIdx.y == 0) {  // (see pyrdown2_kernel)
    clear_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) clear_hist[256] = 0;
  }
  const int ox = blockIdx.x * PD_T, oy = blockIdx.y * PD_T;
  const int sx0 = 2 * ox - 2, sy0 = 2 * oy - 2;
  for (int i = threadIdx.x; i < PD_S * PD_S; i += 256) {
    int ty = i / PD_S, tx = i - ty * PD_S;
    tile[ty][tx] = src[(size_t)reflect101(sy0 + ty, sh_) * sw + reflect101(sx0 + tx, sw)];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PD_S * PD_T; i += 256) {
    int ty = i / PD_T, x = i - ty * PD_T;
    const uint8_t *r = &tile[ty][2 * x];
    hrow[ty][x] = r[0] + 4 * r[1] + 6 * r[2] + 4 * r[3] + r[4];
  }
  __syncthreads();
  const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
  if (ox + x < dw && oy + y < dh) {
    int s = hrow[2 * y][x] + 4 * hrow[2 * y + 1][x] + 6 * hrow[2 * y + 2][x] + 4 * hrow[2 * y + 3][x] + hrow[2 * y + 4][x];
    dst[(size_t)(oy + y) * dw + ox + x] = (uint8_t)((s + 128) >> 8);
  }
}

// Two pyramid levels per launch: a workgroup owns a 16x16 tile of level l+2 and the 32x32 tile of level l+1 under it.  It stages
// the 73x73 footprint of level l, builds the 35x35 piece of level l+1 that the upper tile needs (rounded to u8 exactly as the
// single-level kernel writes it, so every value is the one pyrdown_kernel produces; the halo rows are recomputed by the
// neighbours) and filters again.  Halves the launches of cv::buildOpticalFlowPyramid's chain.
#define PD2_T 8    // (8: 4x the workgroups of 16 — 360 at 752x480 — each a third of the serial staging / filter passes)
#define PD2_M (2 * PD2_T + 3)   // 19: level l+1 piece
#define PD2_S (2 * PD2_M + 3)   // 41: level l footprint
// EQ: `src` is the RAW image and level l is its histogram-equalised version: the workgroup rebuilds the LUT from the finished
// histogram (as equalize_kernel does), maps the footprint while staging it, writes the 64x64 piece of level 0 it owns and clears
// the histogram when it is the last to arrive: cv::equalizeHist's second half and two pyramid levels in one launch.
template <bool EQ>
__global__ void __launch_bounds__(256) pyrdown2_kernel(const uint8_t *__restrict__ src, int sw, int sh, uint8_t *__restrict__ mid,
                                                       int mw, int mh, uint8_t *__restrict__ dst, int dw, int dh,
                                                       unsigned *__restrict__ hist, uint8_t *__restrict__ lvl0, int clear_hist) {
  __shared__ uint8_t t0[PD2_S][PD2_S + 3];
  __shared__ int h0[PD2_S][PD2_M + 1];
  __shared__ uint8_t t1[PD2_M][PD2_M + 1];
  __shared__ int h1[PD2_M][PD2_T + 1];
  __shared__ unsigned cdf[256];
  __shared__ uint8_t lut[256];
  __shared__ bool last_block;
  const int ox = blockIdx.x * PD2_T, oy = blockIdx.y * PD2_T;   // level l+2 tile origin
  // level l+1 rows / columns held in t1: [my0, my0 + PD2_M), clamped at 0 (reads below 0 reflect to 1, 2: inside)
  const int my0 = max(2 * oy - 2, 0), mx0 = max(2 * ox - 2, 0);
  const int sy0 = max(2 * my0 - 2, 0), sx0 = max(2 * mx0 - 2, 0);  // level l origin of t0, same rule
  if (EQ) equalize_lut_256(hist, sw * sh, cdf, lut);  // the LUT of cv::equalizeHist (equalize_lut.hpp)
  if ((sw & 3) == 0) {
    // rows of the footprint as aligned 4-byte words (level widths that are multiples of 4: every word lies inside the row or past its
    // end, and the 64-pixel level-0 piece starts on a word): a quarter of the load / store instructions of the byte loop below
    const int xw0 = sx0 & ~3;
    const int nw = (sx0 + PD2_S - xw0 + 3) >> 2;
    for (int i = threadIdx.x; i < PD2_S * nw; i += 256) {
      const int ty = i / nw, w = i - ty * nw;
      const int Yr = sy0 + ty, Y = min(Yr, sh - 1);
      const int X0 = xw0 + 4 * w;
      unsigned q;
      if (X0 < sw) {
        q = *reinterpret_cast<const unsigned *>(src + (size_t)Y * sw + X0);
      } else {
        q = src[(size_t)Y * sw + sw - 1] * 0x01010101u;  // past the right edge: the last pixel
      }
      if (EQ) {
        q = (unsigned)lut[q & 255] | ((unsigned)lut[(q >> 8) & 255] << 8) | ((unsigned)lut[(q >> 16) & 255] << 16) | ((unsigned)lut[q >> 24] << 24);
        // level 0 rows [4 oy, 4 oy + 64) x columns [4 ox, 4 ox + 64) belong to this workgroup
        if (Yr < sh && X0 < sw && Yr >= 4 * oy && Yr < 4 * oy + 4 * PD2_T && X0 >= 4 * ox && X0 < 4 * ox + 4 * PD2_T)
          *reinterpret_cast<unsigned *>(lvl0 + (size_t)Yr * sw + X0) = q;
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int tx = X0 + b - sx0;
        if (tx >= 0 && tx < PD2_S) t0[ty][tx] = (uint8_t)(q >> (8 * b));
      }
    }
  } else {
    for (int i = threadIdx.x; i < PD2_S * PD2_S; i += 256) {
      const int ty = i / PD2_S, tx = i - ty * PD2_S;
      const int Yr = sy0 + ty, Xr = sx0 + tx;
      const int Y = min(Yr, sh - 1), X = min(Xr, sw - 1);
      uint8_t v = src[(size_t)Y * sw + X];
      if (EQ) {
        v = lut[v];
        // level 0 rows [4 oy, 4 oy + 64) x columns [4 ox, 4 ox + 64) belong to this workgroup
        if (Yr < sh && Xr < sw && Yr >= 4 * oy && Yr < 4 * oy + 4 * PD2_T && Xr >= 4 * ox && Xr < 4 * ox + 4 * PD2_T) lvl0[(size_t)Yr * sw + Xr] = v;
      }
      t0[ty][tx] = v;
    }
  }
  __syncthreads();
  // level l -> l+1, horizontal then vertical; t1[y][x] = level l+1 pixel (my0 + y, mx0 + x)
  for (int i = threadIdx.x; i < PD2_S * PD2_M; i += 256) {
    const int ty = i / PD2_M, x = i - ty * PD2_M;
    const int X = mx0 + x;
    int v = 0;
    if (X < mw && sy0 + ty < sh) {
      const uint8_t *r = t0[ty];
      const int c = 2 * X;
      v = r[reflect101(c - 2, sw) - sx0] + 4 * r[reflect101(c - 1, sw) - sx0] + 6 * r[reflect101(c, sw) - sx0] +
          4 * r[reflect101(c + 1, sw) - sx0] + r[reflect101(c + 2, sw) - sx0];
    }
    h0[ty][x] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PD2_M * PD2_M; i += 256) {
    const int y = i / PD2_M, x = i - y * PD2_M;
    const int Y = my0 + y, X = mx0 + x;
    if (Y < mh && X < mw) {
      const int c = 2 * Y;
      const int s = h0[reflect101(c - 2, sh) - sy0][x] + 4 * h0[reflect101(c - 1, sh) - sy0][x] + 6 * h0[reflect101(c, sh) - sy0][x] +
                    4 * h0[reflect101(c + 1, sh) - sy0][x] + h0[reflect101(c + 2, sh) - sy0][x];
      const uint8_t px = (uint8_t)((s + 128) >> 8);
      t1[y][x] = px;
      // this workgroup owns level l+1 rows [2 oy, 2 oy + 32) x columns [2 ox, 2 ox + 32)
      if (Y >= 2 * oy && Y < 2 * oy + 2 * PD2_T && X >= 2 * ox && X < 2 * ox + 2 * PD2_T) mid[(size_t)Y * mw + X] = px;
    }
  }
  __syncthreads();
  // level l+1 -> l+2
  for (int i = threadIdx.x; i < PD2_M * PD2_T; i += 256) {
    const int ty = i / PD2_T, x = i - ty * PD2_T;
    const int X = ox + x;
    int v = 0;
    if (X < dw && my0 + ty < mh) {
      const uint8_t *r = t1[ty];
      const int c = 2 * X;
      v = r[reflect101(c - 2, mw) - mx0] + 4 * r[reflect101(c - 1, mw) - mx0] + 6 * r[reflect101(c, mw) - mx0] +
          4 * r[reflect101(c + 1, mw) - mx0] + r[reflect101(c + 2, mw) - mx0];
    }
    h1[ty][x] = v;
  }
  __syncthreads();
  if (threadIdx.x < PD2_T * PD2_T) {
    const int x = threadIdx.x % PD2_T, y = threadIdx.x / PD2_T;
    const int X = ox + x, Y = oy + y;
    if (X < dw && Y < dh) {
      const int c = 2 * Y;
      const int s = h1[reflect101(c - 2, mh) - my0][x] + 4 * h1[reflect101(c - 1, mh) - my0][x] + 6 * h1[reflect101(c, mh) - my0][x] +
                    4 * h1[reflect101(c + 1, mh) - my0][x] + h1[reflect101(c + 2, mh) - my0][x];
      dst[(size_t)Y * dw + X] = (uint8_t)((s + 128) >> 8);
    }
  }
  if (!EQ && clear_hist && blockIdx.x == 0 && blockIdx.y == 0) {  // the histogram the previous launch equalised with: zero for the next frame
    hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) hist[256] = 0;
  }
  if (EQ && clear_hist) {  // no further pyramid launch to do it: every workgroup has consumed the histogram (LUT built behind a
                           // barrier), the last one to arrive clears it (an agent-scope fence per workgroup)
    const int t = threadIdx.x;
    if (t == 0) {
      __threadfence();
      last_block = atomicAdd(&hist[256], 1u) == gridDim.x * gridDim.y - 1;
    }
    __syncthreads();
    if (last_block) {
      hist[t] = 0;
      if (t == 0) hist[256] = 0;
    }
  }
}

// ------------------------------------------------------------------------------------------ K3
#define LK_MAXWIN 21             // 15 is the reference's setting and the parity value (TrackKLT.h:143-144); SURVEY D1 / the north star name 21
#define LK_TT (LK_MAXWIN + 3)   // template footprint incl. Scharr halo and bilinear +1
#define LK_JT 32                // search tile edge
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))


// One workgroup of four waves per point; all pyramid levels, coarse to fine.  One window pixel per lane (15 x 15 = 225 of the 256
// lanes); windows of more than 256 pixels (17 x 17 .. 21 x 21, round 4) give the lanes a second pixel each, summed in a wave
// reduction of its own (a wave's sum of one pixel per lane stays below 2^31; of two it would not), and nothing changes for 15 x 15.
// The mismatch sums are exact integers: reduced inside each wave on 32-bit registers (|diff| <= 255 * 32 and |Ix|, |Iy| <=
// 16 * 255 — Scharr on 8-bit pixels — so a pixel's product stays below 2^25 and a wave's sum below 2^31), the four wave sums meet in
// LDS (two slots, alternating by iteration: one barrier per iteration) and every lane adds them as 64-bit integers, so every lane
// takes the same float step.  (Round 2: one wave per point with four pixels per lane took 52 us per launch, this form 45 — the
// launch lasts as long as its slowest point, up to 30 iterations on each of 5 levels.)
#define LK4_WAVES 4
typedef short lk_short2 __attribute__((ext_vector_type(2)));
typedef unsigned __attribute__((aligned(2))) lk_u32a2;  // a 32-bit LDS read at a 16-bit boundary (gfx950 reads LDS unaligned)
// V: 1 = the lean iteration (below; windows of up to 256 pixels), what launch_lk picks; 0 = the loop of rounds 2-4, kept for the
// comparison (tools/lk_exp.py: same bits, 80 -> 73 us per launch at workload C).
// (Round 6, measured and not kept: the window as a compile-time constant — every index of the level set-up a multiply-shift instead of
// a division sequence — 73.8 us against 72.7 with the run-time window at workload C: the launch is bound by the dependent operations of
// its slowest point's iterations, not by the set-up's instruction count; see also lk_ahead_kernel.)
template <int V>
__global__ void __launch_bounds__(64 * LK4_WAVES) lk_kernel(PyrDesc prev, PyrDesc cur, int n, const float *__restrict__ pts0,
                                                            const float *__restrict__ pts1_init, float *__restrict__ pts1,
                                                            uint8_t *__restrict__ status,
                                                            int *__restrict__ iters_out, int win, int max_iters, float eps, CamK K,
                                                            float *__restrict__ n0, float *__restrict__ n1) {
  __shared__ uint8_t ttile[LK_TT][LK_TT + 2];
  __shared__ short tdx[LK_TT - 2][LK_TT - 2], tdy[LK_TT - 2][LK_TT - 2];
  __shared__ uint8_t jtile[LK_JT][LK_JT];
  __shared__ unsigned short jt16[LK_JT][LK_JT];  // (lean iteration: the search tile as 16-bit pixels, two of them per 32-bit read)
  __shared__ int part[2][LK4_WAVES][2], part2[2][LK4_WAVES][2];
  __shared__ int partA[LK4_WAVES][3], partA2[LK4_WAVES][3];
  const int pt = blockIdx.x;
  if (pt >= n) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float half = (win - 1) * 0.5f;
  const float ec = fminf(fmaxf(eps, 0.f), 10.f);
  const double eps2 = (double)ec * (double)ec;
  const int npx = win * win;
  const float px0 = pts0[2 * pt], py0 = pts0[2 * pt + 1];
  const float nx0 = pts1_init[2 * pt], ny0 = pts1_init[2 * pt + 1];
  const int maxLevel = prev.levels - 1;
  float nextx = nx0, nexty = ny0;
  int st = 1, iters = 0, slot = 0;
  const bool own = tid < npx;
  const int wy = own ? tid / win : 0, wx = own ? tid - wy * win : 0;
  const bool two = npx > 64 * LK4_WAVES;  // (uniform) a second window pixel per lane
  const bool own2 = two && tid + 64 * LK4_WAVES < npx;
  const int wy2 = own2 ? (tid + 64 * LK4_WAVES) / win : 0, wx2 = own2 ? tid + 64 * LK4_WAVES - wy2 * win : 0;
  constexpr bool LEAN = (V & 1) != 0;

  for (int level = maxLevel; level >= 0; --level) {
    const float sc = 1.f / (float)(1 << level);
    float prevx = px0 * sc, prevy = py0 * sc;
    if (level == maxLevel) {
      nextx = nx0 * sc;
      nexty = ny0 * sc;
    } else {
      nextx = nextx * 2.f;
      nexty = nexty * 2.f;
    }
    const int cols = prev.w[level], rows = prev.h[level];
    const uint8_t *I = prev.base + prev.off[level];
    const uint8_t *J = cur.base + cur.off[level];
    prevx -= half;
    prevy -= half;
    const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
    if (ipx < -win || ipx >= cols || ipy < -win || ipy >= rows) {
      if (level == 0) st = 0;
      continue;
    }
    const int tt = win + 3;
    __syncthreads();
    for (int i = tid; i < tt * tt; i += 64 * LK4_WAVES) {
      int ty = i / tt, tx = i - ty * tt;
      ttile[ty][tx] = I[(size_t)reflect101(ipy - 1 + ty, rows) * cols + reflect101(ipx - 1 + tx, cols)];
    }
    __syncthreads();
    const int td = win + 1;
    for (int i = tid; i < td * td; i += 64 * LK4_WAVES) {
      int y = i / td, x = i - y * td;
      int X = ipx + x, Y = ipy + y;
      int dx = 0, dy = 0;
      if (X >= 0 && Y >= 0 && X < cols && Y < rows) {  // derivative plane has a CONSTANT(0) border
        const uint8_t *r0 = &ttile[y][x], *r1 = &ttile[y + 1][x], *r2 = &ttile[y + 2][x];
        int t0m = (r0[0] + r2[0]) * 3 + r1[0] * 10, t0p = (r0[2] + r2[2]) * 3 + r1[2] * 10;
        int t1m = r2[0] - r0[0], t1c = r2[1] - r0[1], t1p = r2[2] - r0[2];
        dx = t0p - t0m;
        dy = (t1m + t1p) * 3 + t1c * 10;
      }
      tdx[y][x] = (short)dx;
      tdy[y][x] = (short)dy;
    }
    __syncthreads();
    float a = prevx - ipx, b = prevy - ipy;
    int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
    int iw10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    int Iv = 0, Ix = 0, Iy = 0, Iv2 = 0, Ix2 = 0, Iy2 = 0;
    if (own) {
      Iv = DESCALE(ttile[wy + 1][wx + 1] * iw00 + ttile[wy + 1][wx + 2] * iw01 + ttile[wy + 2][wx + 1] * iw10 + ttile[wy + 2][wx + 2] * iw11,
                   W_BITS - 5);
      Ix = DESCALE(tdx[wy][wx] * iw00 + tdx[wy][wx + 1] * iw01 + tdx[wy + 1][wx] * iw10 + tdx[wy + 1][wx + 1] * iw11, W_BITS);
      Iy = DESCALE(tdy[wy][wx] * iw00 + tdy[wy][wx + 1] * iw01 + tdy[wy + 1][wx] * iw10 + tdy[wy + 1][wx + 1] * iw11, W_BITS);
    }
    if (own2) {
      Iv2 = DESCALE(ttile[wy2 + 1][wx2 + 1] * iw00 + ttile[wy2 + 1][wx2 + 2] * iw01 + ttile[wy2 + 2][wx2 + 1] * iw10 + ttile[wy2 + 2][wx2 + 2] * iw11,
                    W_BITS - 5);
      Ix2 = DESCALE(tdx[wy2][wx2] * iw00 + tdx[wy2][wx2 + 1] * iw01 + tdx[wy2 + 1][wx2] * iw10 + tdx[wy2 + 1][wx2 + 1] * iw11, W_BITS);
      Iy2 = DESCALE(tdy[wy2][wx2] * iw00 + tdy[wy2][wx2 + 1] * iw01 + tdy[wy2 + 1][wx2] * iw10 + tdy[wy2 + 1][wx2 + 1] * iw11, W_BITS);
    }
    {  // |Ix|, |Iy| <= 4080: a product < 2^24, a wave's sum < 2^30
      const int a11 = wave_sum_i32(Ix * Ix), a12 = wave_sum_i32(Ix * Iy), a22 = wave_sum_i32(Iy * Iy);
      if (lane == 0) partA[wave][0] = a11, partA[wave][1] = a12, partA[wave][2] = a22;
      if (two) {
        const int c11 = wave_sum_i32(Ix2 * Ix2), c12 = wave_sum_i32(Ix2 * Iy2), c22 = wave_sum_i32(Iy2 * Iy2);
        if (lane == 0) partA2[wave][0] = c11, partA2[wave][1] = c12, partA2[wave][2] = c22;
      }
    }
    __syncthreads();
    double sA11 = 0.0, sA12 = 0.0, sA22 = 0.0;  // (exact, see the mismatch sums below)
#pragma unroll
    for (int w = 0; w < LK4_WAVES; ++w) sA11 += (double)partA[w][0], sA12 += (double)partA[w][1], sA22 += (double)partA[w][2];
    if (two)
#pragma unroll
      for (int w = 0; w < LK4_WAVES; ++w) sA11 += (double)partA2[w][0], sA12 += (double)partA2[w][1], sA22 += (double)partA2[w][2];
    const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
    if (minEig < 1e-4f || D < 1.1920929e-07f) {
      if (level == 0) st = 0;
      continue;
    }
    D = 1.f / D;
    nextx -= half;
    nexty -= half;
    float pdx = 0.f, pdy = 0.f;
    float outx = nextx + half, outy = nexty + half;
    const int jc = cur.w[level], jr = cur.h[level];
    int jx0 = 0, jy0 = 0;
    bool have_tile = false;
    if (LEAN && !two) {
      // The iteration with as few instructions on the wave as the arithmetic allows (a lone wave retires one instruction per ~7
      // cycles whatever it is: the launch lasts as long as its slowest point's instruction count).  Same values, bit for bit:
      //  * one range test per iteration: the window position against the part of the tile's usable range that lies inside the
      //    image's (fixed when the tile is loaded); the slow path behind it tells "left the image" from "needs another tile";
      //  * the fractions from floorf's own result ((float)(int)floorf(x) == floorf(x)); the bilinear weights scaled by 2^14 before
      //    the product instead of after it (a power of two commutes with the rounding), rounded to nearest-even by adding 2^23;
      //  * the tile holds 16-bit pixels: one 32-bit read per row fetches the pixel pair, two v_dot2_i32_i16 form the interpolation
      //    sum (the fourth weight can be -1: signed);
      //  * (double)ddx * ddx + (double)ddy * ddy as one fma on the exact product (both products are exact in double);
      //  * |(double)s| < 0.01 for a float s is |s| <= 0.01f (0.01 is not a float; 0.01f is the largest float below it); the j > 0
      //    of that test is an infinite previous step.
      int lo_x = -(1 << 28), lo_y = lo_x;
      unsigned span_x = 0, span_y = 0;
      const int lane_off = wy * LK_JT + wx;
      float pvx = __builtin_inff(), pvy = __builtin_inff(), ddx = 0.f, ddy = 0.f;
      bool osc = false;
      for (int j = 0; j < max_iters; ++j) {
        const float fx = floorf(nextx), fy = floorf(nexty);
        const int inx = (int)fx, iny = (int)fy;
        if (((unsigned)(inx - lo_x) > span_x) | ((unsigned)(iny - lo_y) > span_y)) {
          if (inx < -win || inx >= jc || iny < -win || iny >= jr) {
            if (level == 0) st = 0;
            break;
          }
          jx0 = inx - (LK_JT - win - 1) / 2;
          jy0 = iny - (LK_JT - win - 1) / 2;
          __syncthreads();
          {
            const int r = tid >> 3, hx = (tid & 7) * 4;
            const int Y = reflect101(jy0 + r, jr);
#pragma unroll
            for (int c = 0; c < 4; ++c) jt16[r][hx + c] = J[(size_t)Y * jc + reflect101(jx0 + hx + c, jc)];
          }
          __syncthreads();
          lo_x = max(jx0, -win), lo_y = max(jy0, -win);
          span_x = (unsigned)(min(jx0 + LK_JT - win - 1, jc - 1) - lo_x);
          span_y = (unsigned)(min(jy0 + LK_JT - win - 1, jr - 1) - lo_y);
        }
        ++iters;
        const float fa = nextx - fx, fb = nexty - fy;
        const float sa = (1.f - fa) * 16384.f, sb = 1.f - fb, la = fa * 16384.f;
        const unsigned u00 = __float_as_uint(sa * sb + 8388608.f), u01 = __float_as_uint(la * sb + 8388608.f),
                       u10 = __float_as_uint(sa * fb + 8388608.f);
        const unsigned u11 = (16384u + 3u * 0x4B000000u) - u00 - u01 - u10;  // (the biases of the three cancel: the weight itself)
        const unsigned w0 = __builtin_amdgcn_perm(u01, u00, 0x05040100), w1 = __builtin_amdgcn_perm(u11, u10, 0x05040100);
        int pb1 = 0, pb2 = 0;
        if (own) {
          const unsigned short *p = &jt16[0][0] + ((iny - jy0) * LK_JT + (inx - jx0) + lane_off);
          const unsigned r0 = *(const lk_u32a2 *)p, r1 = *(const lk_u32a2 *)(p + LK_JT);
          int v = __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_short2, r0), __builtin_bit_cast(lk_short2, w0), 1 << (W_BITS - 5 - 1), false);
          v = __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_short2, r1), __builtin_bit_cast(lk_short2, w1), v, false);
          const int diff = (v >> (W_BITS - 5)) - Iv;
          pb1 = diff * Ix;
          pb2 = diff * Iy;
        }
        pb1 = wave_sum_i32_lane63(pb1);
        pb2 = wave_sum_i32_lane63(pb2);
        if (lane == 63) part[slot][wave][0] = pb1, part[slot][wave][1] = pb2;  // (the lane the reduction ends in: no broadcast)
        __syncthreads();
        double sb1 = 0.0, sb2 = 0.0;
#pragma unroll
        for (int w = 0; w < LK4_WAVES; ++w) sb1 += (double)part[slot][w][0], sb2 += (double)part[slot][w][1];
        slot ^= 1;
        const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
        ddx = (A12 * b2 - A22 * b1) * D;
        ddy = (A12 * b1 - A11 * b2) * D;
        nextx += ddx;
        nexty += ddy;
        if (fma((double)ddx, (double)ddx, (double)ddy * (double)ddy) <= eps2) break;
        if (fabsf(ddx + pvx) <= 0.01f && fabsf(ddy + pvy) <= 0.01f) {
          osc = true;
          break;
        }
        pvx = ddx;
        pvy = ddy;
      }
      outx = nextx + half;
      outy = nexty + half;
      if (osc) {
        outx -= ddx * 0.5f;
        outy -= ddy * 0.5f;
      }
      nextx = outx;
      nexty = outy;
      continue;
    }
    for (int j = 0; j < max_iters; ++j) {
      const int inx = (int)floorf(nextx), iny = (int)floorf(nexty);
      if (inx < -win || inx >= jc || iny < -win || iny >= jr) {
        if (level == 0) st = 0;
        break;
      }
      ++iters;
      if (!have_tile || inx < jx0 || iny < jy0 || inx + win + 1 > jx0 + LK_JT || iny + win + 1 > jy0 + LK_JT) {
        jx0 = inx - (LK_JT - win - 1) / 2;
        jy0 = iny - (LK_JT - win - 1) / 2;
        __syncthreads();
        {
          const int r = tid >> 3, hx = (tid & 7) * 4;
          const int Y = reflect101(jy0 + r, jr);
#pragma unroll
          for (int c = 0; c < 4; ++c) jtile[r][hx + c] = J[(size_t)Y * jc + reflect101(jx0 + hx + c, jc)];
        }
        __syncthreads();
        have_tile = true;
      }
      a = nextx - inx;
      b = nexty - iny;
      iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
      iw01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
      iw10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
      iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      int pb1 = 0, pb2 = 0;
      if (own) {
        const int x = inx - jx0 + wx, y = iny - jy0 + wy;
        const int diff =
            DESCALE(jtile[y][x] * iw00 + jtile[y][x + 1] * iw01 + jtile[y + 1][x] * iw10 + jtile[y + 1][x + 1] * iw11, W_BITS - 5) - Iv;
        pb1 = diff * Ix;
        pb2 = diff * Iy;
      }
      pb1 = wave_sum_i32(pb1);  // |diff| <= 8160, |Ix| <= 4080: 64 products < 2^31
      pb2 = wave_sum_i32(pb2);
      if (lane == 0) part[slot][wave][0] = pb1, part[slot][wave][1] = pb2;
      if (two) {
        int qb1 = 0, qb2 = 0;
        if (own2) {
          const int x = inx - jx0 + wx2, y = iny - jy0 + wy2;
          const int diff =
              DESCALE(jtile[y][x] * iw00 + jtile[y][x + 1] * iw01 + jtile[y + 1][x] * iw10 + jtile[y + 1][x + 1] * iw11, W_BITS - 5) - Iv2;
          qb1 = diff * Ix2;
          qb2 = diff * Iy2;
        }
        qb1 = wave_sum_i32(qb1);
        qb2 = wave_sum_i32(qb2);
        if (lane == 0) part2[slot][wave][0] = qb1, part2[slot][wave][1] = qb2;
      }
      __syncthreads();
      // the four wave sums (|.| < 2^31 each) are added as doubles: exact (the total stays below 2^33), and (float) of that double
      // rounds once, as (float) of the 64-bit integer does — without the scalar-unit sequence an int64 -> float conversion compiles to
      double sb1 = 0.0, sb2 = 0.0;
#pragma unroll
      for (int w = 0; w < LK4_WAVES; ++w) sb1 += (double)part[slot][w][0], sb2 += (double)part[slot][w][1];
      if (two)
#pragma unroll
        for (int w = 0; w < LK4_WAVES; ++w) sb1 += (double)part2[slot][w][0], sb2 += (double)part2[slot][w][1];
      slot ^= 1;
      const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
      const float ddx = (A12 * b2 - A22 * b1) * D;
      const float ddy = (A12 * b1 - A11 * b2) * D;
      nextx += ddx;
      nexty += ddy;
      outx = nextx + half;
      outy = nexty + half;
      if ((double)ddx * ddx + (double)ddy * ddy <= eps2) break;
      if (j > 0 && fabs((double)(ddx + pdx)) < 0.01 && fabs((double)(ddy + pdy)) < 0.01) {
        outx -= ddx * 0.5f;
        outy -= ddy * 0.5f;
        break;
      }
      pdx = ddx;
      pdy = ddy;
    }
    nextx = outx;
    nexty = outy;
  }
  if (tid == 0) {
    pts1[2 * pt] = nextx;
    pts1[2 * pt + 1] = nexty;
    status[pt] = (uint8_t)st;
    if (iters_out) iters_out[pt] = iters;
  }
  if (n0 && tid < 2) {
    float xn, yn;
    undistort_radtan(K.v, tid == 0 ? px0 : nextx, tid == 0 ? py0 : nexty, xn, yn);
    float *dst = tid == 0 ? n0 : n1;
    dst[2 * pt] = xn;
    dst[2 * pt + 1] = yn;
  }
}

// lk_kernel<1>'s arithmetic with another schedule of its memory traffic (round 6; windows of up to 256 pixels) —
// AN EXPERIMENT, selected by PLV_KNOB_LK_AHEAD, measured NO FASTER (tools/lk_exp.py, workload C, 369 points: 74.8 us per launch against
// 72.7 for lk_kernel<1>; with one iteration per level 28.7 against 27.0): a level's set-up is not waiting for memory.  The launch is as
// long as its slowest point's chain of dependent operations — 82 iterations of ~0.56 us (wave reduction by DPP, the four partial sums
// through LDS and a barrier, the 2 x 2 solve in float) + ~3 us per level + ~12 us of launch, dispatch and the closing undistortion.
// The launch lasts as long as its slowest point, and that point's chain was, per pyramid level: a global read of the template tile,
// barriers, Scharr planes and the normal matrix, then a global read of the search tile before the first iteration — two dependent
// memory latencies (~1.5 us each) on each of five levels.  Here
//  * the template side of EVERY level is done before the first iteration: the tiles of all levels are requested together (one
//    latency), then Scharr planes, the lanes' template values (kept in LDS per level) and the normal matrices follow from LDS alone;
//    the top level's search tile is requested with them;
//  * while a level iterates, the search tile of the level below is on its way into registers (requested around twice the position
//    the level started from) and is laid into the second tile buffer when the level ends; the level below finds it in place when
//    its start lies inside it (the usual case: a level moves the estimate by a pixel or two) and reads its own tile as before when not.
//    The barrier inside the iteration waits for LDS only (s_waitcnt lgkmcnt(0)): a __syncthreads() would wait for the tile as well.
// Every value that reaches the arithmetic is the one lk_kernel<1> reads: same bits (tools/lk_exp.py, tests/test_gpu_frontend.py).
__device__ __forceinline__ void lk_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ void __launch_bounds__(64 * LK4_WAVES) lk_ahead_kernel(PyrDesc prev, PyrDesc cur, int n, const float *__restrict__ pts0,
                                                                  const float *__restrict__ pts1_init, float *__restrict__ pts1,
                                                                  uint8_t *__restrict__ status, int *__restrict__ iters_out, int win,
                                                                  int max_iters, float eps, CamK K, float *__restrict__ n0,
                                                                  float *__restrict__ n1) {
  __shared__ uint8_t tt_all[PLV_MAX_LEVELS][LK_TT][LK_TT + 2];
  __shared__ short tdx[LK_TT - 2][LK_TT - 2], tdy[LK_TT - 2][LK_TT - 2];
  __shared__ int2 tpl[PLV_MAX_LEVELS][64 * LK4_WAVES];  // per level and lane: x = Iv | Ix << 16, y = Iy
  __shared__ int partA_all[PLV_MAX_LEVELS][LK4_WAVES][3];
  __shared__ unsigned short jtb[2][LK_JT][LK_JT];
  __shared__ int part[2][LK4_WAVES][2];
  const int pt = blockIdx.x;
  if (pt >= n) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float half = (win - 1) * 0.5f;
  const float ec = fminf(fmaxf(eps, 0.f), 10.f);
  const double eps2 = (double)ec * (double)ec;
  const int npx = win * win;
  const float px0 = pts0[2 * pt], py0 = pts0[2 * pt + 1];
  const float nx0 = pts1_init[2 * pt], ny0 = pts1_init[2 * pt + 1];
  const int maxLevel = prev.levels - 1;
  const bool own = tid < npx;
  const int wy = own ? tid / win : 0, wx = own ? tid - wy * win : 0;
  const int tt = win + 3, td = win + 1;
  const int tile_r = tid >> 3, tile_hx = (tid & 7) * 4;  // this thread's four pixels of a 32 x 32 search tile
  const int jpad = (LK_JT - win - 1) / 2;

  // ---- every level's template tile (and the top level's search tile) in one round of global reads
  unsigned lvl_in = 0;  // bit l: the template position of level l lies inside the image's margin (the reference's test)
  for (int l = 0; l <= maxLevel; ++l) {
    const float sc = 1.f / (float)(1 << l);
    const int ipx = (int)floorf(px0 * sc - half), ipy = (int)floorf(py0 * sc - half);
    if (!(ipx < -win || ipx >= prev.w[l] || ipy < -win || ipy >= prev.h[l])) lvl_in |= 1u << l;
  }
  for (int i = tid; i < (maxLevel + 1) * tt * tt; i += 64 * LK4_WAVES) {
    const int l = i / (tt * tt), r = i - l * tt * tt;
    if (!((lvl_in >> l) & 1u)) continue;
    const int ty = r / tt, tx = r - ty * tt;
    const float sc = 1.f / (float)(1 << l);
    const int ipx = (int)floorf(px0 * sc - half), ipy = (int)floorf(py0 * sc - half);
    const int cols = prev.w[l], rows = prev.h[l];
    tt_all[l][ty][tx] = (prev.base + prev.off[l])[(size_t)reflect101(ipy - 1 + ty, rows) * cols + reflect101(ipx - 1 + tx, cols)];
  }
  int jb = 0;                     // the tile buffer the current level reads
  bool pf_valid = false;          // a tile for the level about to start lies in jtb[jb] at (pf_x0, pf_y0)
  int pf_x0 = 0, pf_y0 = 0;
  {
    const float sc = 1.f / (float)(1 << maxLevel);
    const int jc = cur.w[maxLevel], jr = cur.h[maxLevel];
    const int inx = (int)floorf(nx0 * sc - half), iny = (int)floorf(ny0 * sc - half);
    if (!(inx < -win || inx >= jc || iny < -win || iny >= jr)) {
      pf_valid = true, pf_x0 = inx - jpad, pf_y0 = iny - jpad;
      const uint8_t *J = cur.base + cur.off[maxLevel];
      const int Y = reflect101(pf_y0 + tile_r, jr);
#pragma unroll
      for (int c = 0; c < 4; ++c) jtb[0][tile_r][tile_hx + c] = J[(size_t)Y * jc + reflect101(pf_x0 + tile_hx + c, jc)];
    }
  }
  __syncthreads();
  // ---- Scharr planes, template values and the parts of the normal matrix, level by level, from LDS
  for (int l = maxLevel; l >= 0; --l) {
    if (!((lvl_in >> l) & 1u)) continue;  // (uniform)
    const float sc = 1.f / (float)(1 << l);
    const float prevx = px0 * sc - half, prevy = py0 * sc - half;
    const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
    const int cols = prev.w[l], rows = prev.h[l];
    for (int i = tid; i < td * td; i += 64 * LK4_WAVES) {
      const int y = i / td, x = i - y * td;
      const int X = ipx + x, Y = ipy + y;
      int dx = 0, dy = 0;
      if (X >= 0 && Y >= 0 && X < cols && Y < rows) {  // derivative plane has a CONSTANT(0) border
        const uint8_t *r0 = &tt_all[l][y][x], *r1 = &tt_all[l][y + 1][x], *r2 = &tt_all[l][y + 2][x];
        const int t0m = (r0[0] + r2[0]) * 3 + r1[0] * 10, t0p = (r0[2] + r2[2]) * 3 + r1[2] * 10;
        const int t1m = r2[0] - r0[0], t1c = r2[1] - r0[1], t1p = r2[2] - r0[2];
        dx = t0p - t0m;
        dy = (t1m + t1p) * 3 + t1c * 10;
      }
      tdx[y][x] = (short)dx;
      tdy[y][x] = (short)dy;
    }
    __syncthreads();
    const float a = prevx - ipx, b = prevy - ipy;
    const int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
    const int iw01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
    const int iw10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
    const int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    int Iv = 0, Ix = 0, Iy = 0;
    if (own) {
      const uint8_t(*T)[LK_TT + 2] = tt_all[l];
      Iv = DESCALE(T[wy + 1][wx + 1] * iw00 + T[wy + 1][wx + 2] * iw01 + T[wy + 2][wx + 1] * iw10 + T[wy + 2][wx + 2] * iw11, W_BITS - 5);
      Ix = DESCALE(tdx[wy][wx] * iw00 + tdx[wy][wx + 1] * iw01 + tdx[wy + 1][wx] * iw10 + tdx[wy + 1][wx + 1] * iw11, W_BITS);
      Iy = DESCALE(tdy[wy][wx] * iw00 + tdy[wy][wx + 1] * iw01 + tdy[wy + 1][wx] * iw10 + tdy[wy + 1][wx + 1] * iw11, W_BITS);
    }
    tpl[l][tid] = make_int2((Iv & 0xffff) | (Ix << 16), Iy);  // (0 <= Iv <= 8160, |Ix|, |Iy| <= 4080)
    const int a11 = wave_sum_i32(Ix * Ix), a12 = wave_sum_i32(Ix * Iy), a22 = wave_sum_i32(Iy * Iy);
    if (lane == 0) partA_all[l][wave][0] = a11, partA_all[l][wave][1] = a12, partA_all[l][wave][2] = a22;
    __syncthreads();  // (the planes are rewritten by the next level; the parts are read below)
  }

  float nextx = nx0, nexty = ny0;
  int st = 1, iters = 0, slot = 0;
  for (int level = maxLevel; level >= 0; --level) {
    const float sc = 1.f / (float)(1 << level);
    if (level == maxLevel) {
      nextx = nx0 * sc;
      nexty = ny0 * sc;
    } else {
      nextx = nextx * 2.f;
      nexty = nexty * 2.f;
    }
    const bool had_pf = pf_valid;  // (a tile requested for THIS level; whatever happens to the level, it is used up)
    const int tx0 = pf_x0, ty0 = pf_y0;
    pf_valid = false;
    if (!((lvl_in >> level) & 1u)) {
      if (level == 0) st = 0;
      continue;
    }
    double sA11 = 0.0, sA12 = 0.0, sA22 = 0.0;
#pragma unroll
    for (int w = 0; w < LK4_WAVES; ++w) sA11 += (double)partA_all[level][w][0], sA12 += (double)partA_all[level][w][1], sA22 += (double)partA_all[level][w][2];
    const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
    if (minEig < 1e-4f || D < 1.1920929e-07f) {
      if (level == 0) st = 0;
      continue;
    }
    D = 1.f / D;
    const int2 tv = tpl[level][tid];
    const int Iv = tv.x & 0xffff, Ix = tv.x >> 16, Iy = tv.y;
    nextx -= half;
    nexty -= half;
    const int jc = cur.w[level], jr = cur.h[level];
    const uint8_t *J = cur.base + cur.off[level];
    int jx0 = 0, jy0 = 0;
    int lo_x = -(1 << 28), lo_y = lo_x;
    unsigned span_x = 0, span_y = 0;
    if (had_pf) {  // the tile requested ahead of this level
      jx0 = tx0, jy0 = ty0;
      lo_x = max(jx0, -win), lo_y = max(jy0, -win);
      span_x = (unsigned)(min(jx0 + LK_JT - win - 1, jc - 1) - lo_x);
      span_y = (unsigned)(min(jy0 + LK_JT - win - 1, jr - 1) - lo_y);
    }
    // the search tile of the level below, around twice this level's start (registers until the level ends)
    unsigned pfv = 0;
    bool pf_next = false;
    int nx0p = 0, ny0p = 0;
    if (level > 0) {
      const int jc1 = cur.w[level - 1], jr1 = cur.h[level - 1];
      const int pinx = (int)floorf((nextx + half) * 2.f - half), piny = (int)floorf((nexty + half) * 2.f - half);
      if (!(pinx < -win || pinx >= jc1 || piny < -win || piny >= jr1)) {
        pf_next = true, nx0p = pinx - jpad, ny0p = piny - jpad;
        const uint8_t *J1 = cur.base + cur.off[level - 1];
        const int Y = reflect101(ny0p + tile_r, jr1);
#pragma unroll
        for (int c = 0; c < 4; ++c) pfv |= (unsigned)J1[(size_t)Y * jc1 + reflect101(nx0p + tile_hx + c, jc1)] << (8 * c);
      }
    }
    const int lane_off = wy * LK_JT + wx;
    float pvx = __builtin_inff(), pvy = __builtin_inff(), ddx = 0.f, ddy = 0.f;
    bool osc = false;
    for (int j = 0; j < max_iters; ++j) {
      const float fx = floorf(nextx), fy = floorf(nexty);
      const int inx = (int)fx, iny = (int)fy;
      if (((unsigned)(inx - lo_x) > span_x) | ((unsigned)(iny - lo_y) > span_y)) {
        if (inx < -win || inx >= jc || iny < -win || iny >= jr) {
          if (level == 0) st = 0;
          break;
        }
        jx0 = inx - jpad;
        jy0 = iny - jpad;
        __syncthreads();
        {
          const int Y = reflect101(jy0 + tile_r, jr);
#pragma unroll
          for (int c = 0; c < 4; ++c) jtb[jb][tile_r][tile_hx + c] = J[(size_t)Y * jc + reflect101(jx0 + tile_hx + c, jc)];
        }
        __syncthreads();
        lo_x = max(jx0, -win), lo_y = max(jy0, -win);
        span_x = (unsigned)(min(jx0 + LK_JT - win - 1, jc - 1) - lo_x);
        span_y = (unsigned)(min(jy0 + LK_JT - win - 1, jr - 1) - lo_y);
      }
      ++iters;
      const float fa = nextx - fx, fb = nexty - fy;
      const float sa = (1.f - fa) * 16384.f, sb = 1.f - fb, la = fa * 16384.f;
      const unsigned u00 = __float_as_uint(sa * sb + 8388608.f), u01 = __float_as_uint(la * sb + 8388608.f),
                     u10 = __float_as_uint(sa * fb + 8388608.f);
      const unsigned u11 = (16384u + 3u * 0x4B000000u) - u00 - u01 - u10;  // (the biases of the three cancel: the weight itself)
      const unsigned w0 = __builtin_amdgcn_perm(u01, u00, 0x05040100), w1 = __builtin_amdgcn_perm(u11, u10, 0x05040100);
      int pb1 = 0, pb2 = 0;
      if (own) {
        const unsigned short *p = &jtb[jb][0][0] + ((iny - jy0) * LK_JT + (inx - jx0) + lane_off);
        const unsigned r0 = *(const lk_u32a2 *)p, r1 = *(const lk_u32a2 *)(p + LK_JT);
        int v = __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_short2, r0), __builtin_bit_cast(lk_short2, w0), 1 << (W_BITS - 5 - 1), false);
        v = __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_short2, r1), __builtin_bit_cast(lk_short2, w1), v, false);
        const int diff = (v >> (W_BITS - 5)) - Iv;
        pb1 = diff * Ix;
        pb2 = diff * Iy;
      }
      pb1 = wave_sum_i32_lane63(pb1);
      pb2 = wave_sum_i32_lane63(pb2);
      if (lane == 63) part[slot][wave][0] = pb1, part[slot][wave][1] = pb2;  // (the lane the reduction ends in: no broadcast)
      lk_lds_barrier();
      double sb1 = 0.0, sb2 = 0.0;
#pragma unroll
      for (int w = 0; w < LK4_WAVES; ++w) sb1 += (double)part[slot][w][0], sb2 += (double)part[slot][w][1];
      slot ^= 1;
      const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
      ddx = (A12 * b2 - A22 * b1) * D;
      ddy = (A12 * b1 - A11 * b2) * D;
      nextx += ddx;
      nexty += ddy;
      if (fma((double)ddx, (double)ddx, (double)ddy * (double)ddy) <= eps2) break;
      if (fabsf(ddx + pvx) <= 0.01f && fabsf(ddy + pvy) <= 0.01f) {
        osc = true;
        break;
      }
      pvx = ddx;
      pvy = ddy;
    }
    float outx = nextx + half, outy = nexty + half;
    if (osc) {
      outx -= ddx * 0.5f;
      outy -= ddy * 0.5f;
    }
    nextx = outx;
    nexty = outy;
    if (level > 0) {  // the tile requested above goes into the other buffer (every lane has left the loop at the same iteration)
      __syncthreads();
      if (pf_next) {
#pragma unroll
        for (int c = 0; c < 4; ++c) jtb[jb ^ 1][tile_r][tile_hx + c] = (unsigned short)((pfv >> (8 * c)) & 255u);
      }
      __syncthreads();
      jb ^= 1;
      pf_valid = pf_next, pf_x0 = nx0p, pf_y0 = ny0p;
    }
  }
  if (tid == 0) {
    pts1[2 * pt] = nextx;
    pts1[2 * pt + 1] = nexty;
    status[pt] = (uint8_t)st;
    if (iters_out) iters_out[pt] = iters;
  }
  if (n0 && tid < 2) {
    float xn, yn;
    undistort_radtan(K.v, tid == 0 ? px0 : nextx, tid == 0 ? py0 : nexty, xn, yn);
    float *dst = tid == 0 ? n0 : n1;
    dst[2 * pt] = xn;
    dst[2 * pt + 1] = yn;
  }
}

// ------------------------------------------------------------------------------------------ K6
__global__ void undistort_kernel(CamK K, int n, const float *__restrict__ uv, float *__restrict__ xy) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  undistort_radtan(K.v, uv[2 * i], uv[2 * i + 1], xy[2 * i], xy[2 * i + 1]);
}
// both point sets of perform_matching in one launch
__global__ void undistort2_kernel(CamK K, int n, const float *__restrict__ uv0, const float *__restrict__ uv1,
                                  float *__restrict__ xy0, float *__restrict__ xy1) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n) return;
  if (i < n)
    undistort_radtan(K.v, uv0[2 * i], uv0[2 * i + 1], xy0[2 * i], xy0[2 * i + 1]);
  else {
    i -= n;
    undistort_radtan(K.v, uv1[2 * i], uv1[2 * i + 1], xy1[2 * i], xy1[2 * i + 1]);
  }
}

// ------------------------------------------------------------------------------------------ K7
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned rng_draw(unsigned seed, unsigned hyp, unsigned t) {
  return hash32(seed ^ hash32(hyp * 0x9E3779B9U + hash32(t + 0x85EBCA6BU)));
}
__device__ __forceinline__ double det3(const double *m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// cos(acos(x) / 3) and the cube root from +, -, *, / and sqrt only (all correctly rounded on the host and on gfx950), so that the
// CPU restatement and the HIP kernel produce the same fundamental-matrix candidates bit for bit; libm's acos / cos / cbrt differ
// in the last place between the two.  cos(acos(x)/3) is the root of 4 t^3 - 3 t = x in [1/2, 1] (monotone there): 64 bisection
// steps.  Near x = -1 the root is double (t = 1/2) and only sqrt(eps) of it is determined — as ill-conditioned as the cubic's own
// roots are there.  The cube root: bit-level first guess, six Newton steps.
__device__ double det_cos_third(double x) {
  double lo = 0.5, hi = 1.0;
  for (int it = 0; it < 64; ++it) {
    const double mid = 0.5 * (lo + hi);
    const double f = ((4.0 * mid) * mid) * mid - 3.0 * mid - x;
    if (f < 0)
      lo = mid;
    else
      hi = mid;
  }
  return 0.5 * (lo + hi);
}
__device__ double det_cbrt(double x) {  // x >= 0
  if (!(x > 0)) return 0.0;
  unsigned long long i;
  memcpy(&i, &x, 8);
  i = i / 3 + 0x2A9F7893782DA1CEull;
  double y;
  memcpy(&y, &i, 8);
  for (int it = 0; it < 6; ++it) y = y - (y * y * y - x) / (3.0 * y * y);
  return y;
}

__device__ int solve_cubic(double c3, double c2, double c1, double c0, double *roots) {
  if (c3 == 0) {
    if (c2 == 0) {
      if (c1 == 0) return 0;
      roots[0] = -c0 / c1;
      return 1;
    }
    double d = c1 * c1 - 4 * c2 * c0;
    if (d < 0) return 0;
    d = sqrt(d);
    double q = 1. / (2 * c2);
    roots[0] = (-c1 - d) * q;
    roots[1] = (-c1 + d) * q;
    return d > 0 ? 2 : 1;
  }
  double a1 = c2 / c3, a2 = c1 / c3, a3 = c0 / c3;
  double Q = (a1 * a1 - 3 * a2) * (1. / 9);
  double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
  double Qcubed = Q * Q * Q;
  double d = Qcubed - R * R;
  if (d > 0) {
    // theta = acos(R / sqrt(Q^3)); the roots are -2 sqrt(Q) cos(theta / 3 + 2 pi k / 3) - a1 / 3
    double xr = R / sqrt(Qcubed);
    xr = xr < -1.0 ? -1.0 : (xr > 1.0 ? 1.0 : xr);
    const double ct = det_cos_third(xr), st = sqrt(1.0 - ct * ct);
    double sqrtQ = sqrt(Q);
    double t0 = -2 * sqrtQ, t2 = a1 * (1. / 3);
    roots[0] = t0 * ct - t2;
    roots[1] = t0 * (-0.5 * ct - 0.8660254037844386 * st) - t2;
    roots[2] = t0 * (-0.5 * ct + 0.8660254037844386 * st) - t2;
    return 3;
  } else if (d == 0) {
    if (R >= 0) {
      roots[0] = -2 * det_cbrt(R) - a1 / 3;
      roots[1] = det_cbrt(R) - a1 / 3;
    } else {
      roots[0] = 2 * det_cbrt(-R) - a1 / 3;
      roots[1] = -det_cbrt(-R) - a1 / 3;
    }
    return 2;
  } else {
    d = sqrt(-d);
    double e = det_cbrt(d + fabs(R));
    if (R > 0) e = -e;
    roots[0] = (e + Q / e) - a1 * (1. / 3);
    return 1;
  }
}

// ---- wave-parallel hypothesis solver -------------------------------------------------------
// One wavefront per hypothesis.  The subset draw, the cubic and the model normalisation are
// lane-uniform (every lane computes the same scalars, all arrays statically indexed: nothing goes
// to scratch); the 7x9 Gauss-Jordan with full pivoting is lane-parallel, lane l < 63 owning
// A[l/9][l%9]: pivot search = DPP max + ballot (lowest lane wins ties = the serial scan order),
// row/column swaps and the elimination = lane shuffles.  Same arithmetic per element as the
// serial restatement in oracle/frontend_oracle.cpp.
struct RansacLds {
  double f1[9], f2[9];
};

template <int CTRL> __device__ __forceinline__ double dpp_max_step(double v) { return fmax(v, dpp_mov_f64<CTRL>(v)); }
__device__ __forceinline__ double wave_max_nonneg_f64(double v) {  // valid for keys >= -1 with max >= 0
  v = dpp_max_step<0x111>(v);
  v = dpp_max_step<0x112>(v);
  v = dpp_max_step<0x114>(v);
  v = dpp_max_step<0x118>(v);
  v = dpp_max_step<0x142>(v);
  v = dpp_max_step<0x143>(v);
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ bool collinear_last7(const double (&x)[7], const double (&y)[7]) {
  bool col = false;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double dx1 = x[j] - x[6], dy1 = y[j] - y[6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (k < j) {
        const double dx2 = x[k] - x[6], dy2 = y[k] - y[6];
        col = col || (fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920929e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)));
      }
  }
  return col;
}

__device__ __forceinline__ float epi_err9(const double (&F)[9], float fx1, float fy1, float fx2, float fy2) {
  double x1 = fx1, y1 = fy1, x2 = fx2, y2 = fy2;
  double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
  double s2 = 1. / (a * a + b * b);
  double d2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6];
  b = F[1] * x2 + F[4] * y2 + F[7];
  c = F[2] * x2 + F[5] * y2 + F[8];
  double s1 = 1. / (a * a + b * b);
  double d1 = x1 * a + y1 * b + c;
  return (float)fmax(d1 * d1 * s1, d2 * d2 * s2);
}
__device__ __forceinline__ float epi_err9(const double (&F)[9], const float *m1, const float *m2, int i) {
  return epi_err9(F, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]);
}

// Models of hypothesis h.  Returns a 3-bit validity mask; model k (root k of the cubic, in
// cv::solveCubic order) is F[k].  All 64 lanes must call this together.
__device__ int wave_models(RansacLds &L, const float *m1, const float *m2, int n, unsigned seed, int h, double (&F)[3][9]) {
  const int lane = threadIdx.x & 63;
  // ---- 7 distinct indices + collinearity test (uniform)
  double x1[7], y1[7], x2[7], y2[7];
  bool ok = false;
  unsigned t = 0;
  for (int attempt = 0; attempt < 16 && !ok; ++attempt) {
    int idx[7] = {-1, -1, -1, -1, -1, -1, -1};
    int cnt = 0;
    if (n == 7) {
#pragma unroll
      for (int q = 0; q < 7; ++q) idx[q] = q;
      cnt = 7;
    } else {
      for (int guard = 0; guard < 64 && cnt < 7; ++guard) {
        const int cand = (int)(((unsigned long long)rng_draw(seed, (unsigned)h, t++) * (unsigned long long)n) >> 32);
        bool dup = false;
#pragma unroll
        for (int q = 0; q < 7; ++q) dup = dup || (q < cnt && idx[q] == cand);
        if (!dup) {
#pragma unroll
          for (int q = 0; q < 7; ++q) idx[q] = (q == cnt) ? cand : idx[q];
          ++cnt;
        }
      }
    }
    if (cnt < 7) continue;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      x1[q] = m1[2 * idx[q]];
      y1[q] = m1[2 * idx[q] + 1];
      x2[q] = m2[2 * idx[q]];
      y2[q] = m2[2 * idx[q] + 1];
    }
    if (n == 7) {
      ok = true;
    } else if (!(collinear_last7(x1, y1) || collinear_last7(x2, y2))) {
      ok = true;
    }
  }
  if (!ok) return 0;
  // ---- this lane's element of the 7x9 constraint matrix
  const int r = lane < 63 ? lane / 9 : 7, c = lane < 63 ? lane - 9 * (lane / 9) : 0;
  double px0 = 0, py0 = 0, px1 = 0, py1 = 0;
#pragma unroll
  for (int q = 0; q < 7; ++q)
    if (q == r) px0 = x1[q], py0 = y1[q], px1 = x2[q], py1 = y2[q];
  double a;
  switch (c) {
    case 0: a = px1 * px0; break;
    case 1: a = px1 * py0; break;
    case 2: a = px1; break;
    case 3: a = py1 * px0; break;
    case 4: a = py1 * py0; break;
    case 5: a = py1; break;
    case 6: a = px0; break;
    case 7: a = py0; break;
    default: a = 1.0; break;
  }
  if (lane == 63) a = 0.0;
  int cp[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const double key = (lane < 63 && r >= i && c >= i) ? fabs(a) : -1.0;
    const double best = wave_max_nonneg_f64(key);
    if (!(best > 1e-14)) return 0;
    const unsigned long long bal = __ballot(key == best);
    const int pl = __ffsll((long long)bal) - 1;
    const int pr = pl / 9, pc = pl - 9 * pr;
    int src = lane;
    if (lane < 63) {
      if (r == i) src = pr * 9 + c;
      else if (r == pr) src = i * 9 + c;
    }
    a = __shfl(a, src, 64);
    src = lane;
    if (lane < 63) {
      if (c == i) src = r * 9 + pc;
      else if (c == pc) src = r * 9 + i;
    }
    a = __shfl(a, src, 64);
    {  // colperm swap (pc dynamic, i static)
      int cpc = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) cpc = (q == pc) ? cp[q] : cpc;
      const int cpi = cp[i];
#pragma unroll
      for (int q = 0; q < 9; ++q) cp[q] = (q == pc) ? cpi : cp[q];
      cp[i] = cpc;
    }
    const double p = __shfl(a, i * 10, 64);
    const double inv = 1.0 / p;
    if (r == i) a *= inv;
    const double f = __shfl(a, lane < 63 ? r * 9 + i : lane, 64);
    const double arow = __shfl(a, lane < 63 ? i * 9 + c : lane, 64);
    if (lane < 63 && r != i && f != 0.0) a -= f * arow;
  }
  // ---- null-space basis through LDS: f1[cp[j]] = j < 7 ? -A[j][7] : (j == 7), f2 likewise with column 8
  {
    const double v7 = __shfl(a, lane < 7 ? lane * 9 + 7 : lane, 64);
    const double v8 = __shfl(a, lane < 7 ? lane * 9 + 8 : lane, 64);
    int cpl = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) cpl = (q == lane) ? cp[q] : cpl;
    __syncthreads();
    if (lane < 9) {
      L.f1[cpl] = lane < 7 ? -v7 : (lane == 7 ? 1.0 : 0.0);
      L.f2[cpl] = lane < 7 ? -v8 : (lane == 8 ? 1.0 : 0.0);
    }
    __syncthreads();
  }
  double f2[9], g[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    f2[q] = L.f2[q];
    g[q] = L.f1[q] - f2[q];
  }
  // det(f2 + lambda g) = c0 + c1 lambda + c2 lambda^2 + c3 lambda^3
  double c0 = det3(f2), c3 = det3(g), c1 = 0, c2 = 0;
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    double m[9], q[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) m[e] = f2[e], q[e] = g[e];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      m[3 * rr + cc] = g[3 * rr + cc];
      q[3 * rr + cc] = f2[3 * rr + cc];
    }
    c1 += det3(m);
    c2 += det3(q);
  }
  double roots[3] = {0, 0, 0};
  const int nroot = solve_cubic(c3, c2, c1, c0, roots);
  int valid = 0;
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    double lambda = roots[kk], mu = 1.;
    const double s = g[8] * lambda + f2[8];
    if (fabs(s) > 2.220446049250313e-16) {
      mu = 1. / s;
      lambda *= mu;
      F[kk][8] = 1.;
    } else
      F[kk][8] = 0.;
    bool finite = true;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      F[kk][e] = g[e] * lambda + f2[e] * mu;
      finite = finite && isfinite(F[kk][e]);
    }
    if (kk < nroot && finite) valid |= 1 << kk;
  }
  return valid;
}

// counts[h*3 + k] = inliers of the k-th VALID model of hypothesis h (compacted like the serial
// loop), -1 beyond.
__global__ void __launch_bounds__(64) ransac_hyp_kernel(const float *__restrict__ m1, const float *__restrict__ m2, int n,
                                                        float t, unsigned seed, int *__restrict__ counts,
                                                        double *__restrict__ models /* [hyp][28]: 3 x 9 + valid mask, or null */) {
  __shared__ RansacLds L;
  const int h = blockIdx.x, lane = threadIdx.x;
  // this lane's points of the inlier count (lane, lane + 64, ..): fetched before the solver starts, so that the three counts below
  // run on registers instead of waiting for memory once per point and model (18 dependent rounds of loads at 340 points)
  constexpr int RH_PL = 8;
  float2 q1[RH_PL], q2[RH_PL];
#pragma unroll
  for (int u = 0; u < RH_PL; ++u) {
    const int i = min(lane + 64 * u, n - 1);
    q1[u] = reinterpret_cast<const float2 *>(m1)[i];
    q2[u] = reinterpret_cast<const float2 *>(m2)[i];
  }
  double F[3][9];
  const int valid = wave_models(L, m1, m2, n, seed, h, F);
  if (models && lane < 28) {  // kept for ransac_select_kernel: the winner's model is read back instead of being solved again
    double v = (double)valid;
#pragma unroll
    for (int e = 0; e < 27; ++e) v = (lane == e) ? F[e / 9][e % 9] : v;
    models[(size_t)h * 28 + lane] = v;
  }
  int slot = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (valid & (1 << k)) {
      int c = 0;
#pragma unroll
      for (int u = 0; u < RH_PL; ++u)
        if (lane + 64 * u < n) c += epi_err9(F[k], q1[u].x, q1[u].y, q2[u].x, q2[u].y) <= t;
      for (int i = lane + 64 * RH_PL; i < n; i += 64) c += epi_err9(F[k], m1, m2, i) <= t;
      c = wave_sum_i32(c);
      if (lane == 0) counts[h * 3 + slot] = c;
      ++slot;
    }
  }
  if (lane == 0)
    for (int k = slot; k < 3; ++k) counts[h * 3 + k] = -1;
}

__device__ int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
  p = fmin(fmax(p, 0.), 1.);
  ep = fmin(fmax(ep, 0.), 1.);
  double num = fmax(1. - p, 2.2250738585072014e-308);
  double denom = 1. - pow(1. - ep, (double)model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// Replays cv::RANSACPointSetRegistrator::run's adaptive loop over the precomputed counts, 64
// hypotheses per pass: hypothesis `it` is processed iff it < niters(best before it), where
// niters = RANSACUpdateNumIters chained from max_iters is monotone in the running best, so the
// processed set is a prefix and the winner is the first occurrence of the prefix maximum.
// Then re-solves the winning hypothesis and writes mask = klt_status & inlier
// (REF: TrackKLT.cpp:876-879).  klt may be null.  info[0] = inliers, info[1] = iterations used.
__global__ void __launch_bounds__(256) ransac_select_kernel(const float *__restrict__ m1, const float *__restrict__ m2, int n,
                                                           float t, double conf, int max_iters, unsigned seed,
                                                           const int *__restrict__ counts, const uint8_t *__restrict__ klt,
                                                           uint8_t *__restrict__ mask, int *__restrict__ info,
                                                           const double *__restrict__ models, const unsigned *__restrict__ mir_src,
                                                           unsigned *__restrict__ mir_dst, int mir_words, uint8_t *__restrict__ mir_mask,
                                                           unsigned *done_word, unsigned done_val) {
  __shared__ RansacLds L;
  __shared__ double sF[9];
  __shared__ int s_kroot;
  // (round 6b) 256 threads when the hypotheses' models are at hand: the replay of the adaptive loop stays with wave 0, the copy for the
  // host and the inlier mask are spread over all four waves (64 threads when the winner has to be solved again: wave_models' barriers)
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  // (perform_matching: this is the last kernel of the call; it copies what lk_kernel left for the host — positions, normalised
  // coordinates, iteration counts — and its own mask into the caller's pinned buffer, so that no copy command follows it)
  if (mir_dst)
    for (int i = tid; i < mir_words; i += nthr) mir_dst[i] = mir_src[i];
  int kroot = -1;
  double Fk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (tid < 64) {
  const int total = n == 7 ? 1 : max_iters;
  int best = 0, bh = -1, bslot = 0, niters = total, used = total;
  bool stop = false;
  for (int base = 0; base < total && !stop; base += 64) {
    const int it = base + lane;
    int c0 = -1, c1 = -1, c2 = -1;
    if (it < total) {
      c0 = counts[it * 3];
      c1 = counts[it * 3 + 1];
      c2 = counts[it * 3 + 2];
    }
    const int lmax = max(c0, max(c1, c2));
    // exclusive prefix maximum over lanes, seeded with the carried best
    int pre = lmax;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int o = __shfl_up(pre, off, 64);
      if (lane >= off) pre = max(pre, o);
    }
    int excl = __shfl_up(pre, 1, 64);
    if (lane == 0) excl = -1;
    excl = max(excl, best);
    // iteration budget in force when hypothesis `it` is reached
    int budget = niters;
    if (excl > 6 && excl > best) budget = min(niters, ransac_update_iters(conf, (double)(n - excl) / n, 7, niters));
    const bool processed = it < total && it < budget;
    const unsigned long long pb = __ballot(processed);
    const int nproc = pb == ~0ULL ? 64 : __ffsll((long long)~pb) - 1;  // processed lanes form a prefix
    // running best over the processed prefix: first lane (and first slot) reaching the prefix maximum
    const int cand = (lane < nproc) ? lmax : -1;
    int cmax = cand;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off, 64));
    if (cmax > max(best, 6)) {
      const unsigned long long wb = __ballot(cand == cmax);
      const int wl = __ffsll((long long)wb) - 1;
      const int w0 = __shfl(c0, wl, 64), w1 = __shfl(c1, wl, 64);
      best = cmax;
      bh = base + wl;
      bslot = (w0 == cmax) ? 0 : ((w1 == cmax) ? 1 : 2);
      niters = min(niters, ransac_update_iters(conf, (double)(n - best) / n, 7, niters));
    }
    if (nproc < 64 || base + 64 >= niters) {
      stop = true;
      used = min(base + nproc, total);
      if (nproc == 64) used = min(max(niters, base + 64), total);
    }
  }
  if (!stop) used = min(niters, total);
  double F[3][9];
  int valid = 0;
  if (bh >= 0) {
    if (models) {  // the models ransac_hyp_kernel solved (same bits as a second solve)
#pragma unroll
      for (int e = 0; e < 27; ++e) F[e / 9][e % 9] = models[(size_t)bh * 28 + e];
      valid = (int)models[(size_t)bh * 28 + 27];
    } else {
      valid = wave_models(L, m1, m2, n, seed, bh, F);
    }
  }
  // slot -> root index
  int seen = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (valid & (1 << k)) {
      if (seen == bslot && kroot < 0) kroot = k;
      ++seen;
    }
  if (lane == 0) {
    info[0] = best;
    info[1] = used;
  }
#pragma unroll
  for (int e = 0; e < 9; ++e) Fk[e] = kroot == 0 ? F[0][e] : (kroot == 1 ? F[1][e] : F[2][e]);
  if (nthr > 64 && lane == 0) {
    s_kroot = kroot;
#pragma unroll
    for (int e = 0; e < 9; ++e) sF[e] = Fk[e];
  }
  }  // (wave 0)
  if (nthr > 64) {
    __syncthreads();
    kroot = s_kroot;
#pragma unroll
    for (int e = 0; e < 9; ++e) Fk[e] = sF[e];
  }
  for (int i = tid; i < n; i += nthr) {
    const uint8_t in = kroot >= 0 ? (epi_err9(Fk, m1, m2, i) <= t) : 0;
    const uint8_t mv = (in && (!klt || klt[i])) ? 1 : 0;
    mask[i] = mv;
    if (mir_mask) mir_mask[i] = mv;
  }
  if (done_word) {  // everything the host reads of this call is written: say so (every thread's stores precede its fence, the word the barrier)
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(done_word, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


// ------------------------------------------------------------------------------------------ CLAHE
// cv::createCLAHE(10.0, 8x8)->apply   REF call sites: TrackKLT.cpp:60-64, TrackLSD.cpp:84-88.
// clahe_lut_kernel: workgroup per tile — LDS histogram, clip + redistribution (batch, then the strided
// residual exactly as OpenCV walks it), inclusive scan, LUT = round(sum * 255 / area).
// clahe_apply_kernel: per pixel bilinear blend of the four neighbouring tile LUTs in float.
// The tile grid must divide the image (752x480, 1280x560 and 1280x720 do); otherwise PLV_E_BADARG.
__global__ void __launch_bounds__(256) clahe_lut_kernel(const uint8_t *__restrict__ src, int w, int tw, int th, int tiles_x,
                                                        int clip, float lut_scale, uint8_t *__restrict__ lut) {
  __shared__ int hist[256];
  __shared__ int scan[256];
  __shared__ int clipped_total;
  const int t = threadIdx.x;
  const int ti = blockIdx.x % tiles_x, tj = blockIdx.x / tiles_x;
  hist[t] = 0;
  if (t == 0) clipped_total = 0;
  __syncthreads();
  for (int i = t; i < tw * th; i += 256) {
    const int y = i / tw, x = i - y * tw;
    atomicAdd(&hist[src[(size_t)(tj * th + y) * w + ti * tw + x]], 1);
  }
  __syncthreads();
  int hv = hist[t];
  if (clip > 0) {
    if (hv > clip) {
      atomicAdd(&clipped_total, hv - clip);
      hv = clip;
    }
    __syncthreads();
    const int clipped = clipped_total;
    const int batch = clipped / 256;
    const int residual = clipped - batch * 256;
    hv += batch;
    if (residual != 0) {
      const int step = max(256 / residual, 1);
      // bins 0, step, 2*step, ... get one more, `residual` of them at most, while the bin index stays < 256
      if (t % step == 0 && t / step < residual) hv += 1;
    }
  }
  scan[t] = hv;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = t >= off ? scan[t - off] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  const int v = __float2int_rn((float)scan[t] * lut_scale);
  lut[(size_t)blockIdx.x * 256 + t] = (uint8_t)min(max(v, 0), 255);
}

__global__ void __launch_bounds__(256) clahe_apply_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int h,
                                                          int tw, int th, int tiles_x, int tiles_y,
                                                          const uint8_t *__restrict__ lut) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
  const float tyf = y * inv_th - 0.5f, txf = x * inv_tw - 0.5f;
  int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
  int ty2 = ty1 + 1, tx2 = tx1 + 1;
  const float ya = tyf - ty1, ya1 = 1.0f - ya, xa = txf - tx1, xa1 = 1.0f - xa;
  ty1 = max(ty1, 0);
  ty2 = min(ty2, tiles_y - 1);
  tx1 = max(tx1, 0);
  tx2 = min(tx2, tiles_x - 1);
  const int v = src[(size_t)y * w + x];
  const float a = lut[(size_t)(ty1 * tiles_x + tx1) * 256 + v], b = lut[(size_t)(ty1 * tiles_x + tx2) * 256 + v];
  const float c = lut[(size_t)(ty2 * tiles_x + tx1) * 256 + v], d = lut[(size_t)(ty2 * tiles_x + tx2) * 256 + v];
  const float res = (a * xa1 + b * xa) * ya1 + (c * xa1 + d * xa) * ya;
  dst[(size_t)y * w + x] = (uint8_t)min(max(__float2int_rn(res), 0), 255);
}

// ========================================================================================== launchers
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

int launch_equalize(plv_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst, int npix, unsigned *d_hist) {
  int blocks = min(256, max(1, cdiv(npix / 16, 256)));
  {
    ProfScope ps(ctx->prof, "hist_kernel", ctx->stream);
    hipLaunchKernelGGL(hist_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, d_src, npix, d_hist, (uint8_t *)nullptr);
  }
  {
    ProfScope ps(ctx->prof, "equalize_kernel", ctx->stream);
    hipLaunchKernelGGL(equalize_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_src, d_dst, npix, d_hist);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_clahe(plv_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst, int w, int h, double clip_limit, int tiles, uint8_t *d_lut) {
  if (w % tiles != 0 || h % tiles != 0) {
    set_last_error("CLAHE: %dx%d is not divisible into %dx%d tiles (the padded variant is not built)", w, h, tiles, tiles);
    return PLV_E_BADARG;
  }
  const int tw = w / tiles, th = h / tiles, area = tw * th;
  int clip = 0;
  if (clip_limit > 0.0) clip = std::max((int)(clip_limit * area / 256), 1);
  {
    ProfScope ps(ctx->prof, "clahe_lut_kernel", ctx->stream);
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles * tiles), dim3(256), 0, ctx->stream, d_src, w, tw, th, tiles, clip,
                       (float)255 / area, d_lut);
  }
  {
    ProfScope ps(ctx->prof, "clahe_apply_kernel", ctx->stream);
    hipLaunchKernelGGL(clahe_apply_kernel, dim3(cdiv(w, 64), cdiv(h, 4)), dim3(256), 0, ctx->stream, d_src, d_dst, w, h, tw, th, tiles,
                       tiles, d_lut);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

// cv::equalizeHist + cv::buildOpticalFlowPyramid from the raw image: histogram, then the equalisation rides on the first
// two-level pyramid launch (needs at least three levels; otherwise the separate kernels)
int launch_equalize_pyramid(plv_ctx *ctx, const uint8_t *d_raw, const PyrDesc &p, unsigned *d_hist, const uint8_t *h_src) {
  const int npix = p.w[0] * p.h[0];
  if (p.levels < 3) {
    if (h_src) PLV_HIP_CHECK(plv::memcpy_async((void *)d_raw, h_src, (size_t)npix, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_equalize(ctx, d_raw, p.base + p.off[0], npix, d_hist);
    return rc ? rc : launch_pyramid(ctx, p);
  }
  {
    ProfScope ps(ctx->prof, "hist_kernel", ctx->stream);
    const int blocks = min(256, max(1, cdiv(npix / 16, 256)));
    // h_src: the image is still in the library's pinned host block — this kernel reads it from there and leaves it in d_raw
    if (h_src)
      hipLaunchKernelGGL(hist_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, h_src, npix, d_hist, const_cast<uint8_t *>(d_raw));
    else
      hipLaunchKernelGGL(hist_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, d_raw, npix, d_hist, (uint8_t *)nullptr);
  }
  // (the line detector's edge kernel, when the tracker feed asked for it: it equalises the raw image itself and so need not wait for
  // the pyramid — its maps reach the library's line worker two launches earlier, and the flow starts when it always did)
  struct Rest {
    plv_ctx *ctx;
    const uint8_t *d_raw;
    const PyrDesc *p;
    unsigned *d_hist;
    bool done;
    int rc;
  } rest{ctx, d_raw, &p, d_hist, false, PLV_OK};
  auto run_rest = [](void *a) -> int {
    Rest &R = *(Rest *)a;
    if (R.done) return R.rc;
    R.done = true;
    plv_ctx *ctx = R.ctx;
    const PyrDesc &p = *R.p;
    {
      ProfScope ps(ctx->prof, "pyrdown2_kernel", ctx->stream);
      dim3 grid(cdiv(p.w[2], PD2_T), cdiv(p.h[2], PD2_T));
      // the histogram is cleared for the next frame by the pyramid launch that follows, or here when there is none
      hipLaunchKernelGGL(pyrdown2_kernel<true>, grid, dim3(256), 0, ctx->stream, R.d_raw, p.w[0], p.h[0], p.base + p.off[1], p.w[1], p.h[1],
                         p.base + p.off[2], p.w[2], p.h[2], R.d_hist, p.base + p.off[0], p.levels > 3 ? 0 : 1);
    }
    if (hipGetLastError() != hipSuccess) return R.rc = PLV_E_DEVICE;
    return R.rc = launch_pyramid(ctx, p, 2, R.d_hist);
  };
  if (ctx->edges_hook) {
    // (the hook launches its edge kernel, then calls after_edges — the pyramid goes onto the stream right behind that kernel — and
    // only then spends its host time on the label kernels, the events and the line worker's job)
    ctx->after_edges = run_rest, ctx->after_edges_arg = &rest;
    ctx->edges_hook(ctx, d_raw, p.w[0], p.h[0], d_hist);
    ctx->after_edges = nullptr, ctx->after_edges_arg = nullptr;
  }
  return run_rest(&rest);
}

int launch_pyramid(plv_ctx *ctx, const PyrDesc &p, int first_level, unsigned *clear_hist) {
  int l = first_level;
  for (; l + 2 < p.levels; l += 2) {  // two levels per launch
    ProfScope ps(ctx->prof, "pyrdown2_kernel", ctx->stream);
    dim3 grid(cdiv(p.w[l + 2], PD2_T), cdiv(p.h[l + 2], PD2_T));
    // (the grid of the upper level also covers the level under it: ceil(w2 / T) * 2 T >= w1 because w2 = (w1 + 1) / 2)
    hipLaunchKernelGGL(pyrdown2_kernel<false>, grid, dim3(256), 0, ctx->stream, p.base + p.off[l], p.w[l], p.h[l], p.base + p.off[l + 1],
                       p.w[l + 1], p.h[l + 1], p.base + p.off[l + 2], p.w[l + 2], p.h[l + 2], clear_hist, (uint8_t *)nullptr, clear_hist ? 1 : 0);
    clear_hist = nullptr;
  }
  for (; l + 1 < p.levels; ++l) {
    ProfScope ps(ctx->prof, "pyrdown_kernel", ctx->stream);
    dim3 grid(cdiv(p.w[l + 1], PD_T), cdiv(p.h[l + 1], PD_T));
    hipLaunchKernelGGL(pyrdown_kernel, grid, dim3(256), 0, ctx->stream, p.base + p.off[l], p.w[l], p.h[l],
                       p.base + p.off[l + 1], p.w[l + 1], p.h[l + 1], clear_hist);
    clear_hist = nullptr;
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_pyrdown(plv_ctx *ctx, const uint8_t *d_src, int sw, int sh, uint8_t *d_dst, int dw, int dh) {
  ProfScope ps(ctx->prof, "pyrdown_kernel", ctx->stream);
  hipLaunchKernelGGL(pyrdown_kernel, dim3(cdiv(dw, PD_T), cdiv(dh, PD_T)), dim3(256), 0, ctx->stream, d_src, sw, sh, d_dst, dw, dh, (unsigned *)nullptr);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_lk(plv_ctx *ctx, const PyrDesc &prev, const PyrDesc &cur, int n, const float *d_pts0, float *d_pts1,
              uint8_t *d_status, int *d_iters, int win, int max_iters, float eps, const CamK *K, float *d_n0, float *d_n1,
              const float *pts1_init) {
  if (win > LK_MAXWIN || win < 3 || (win & 1) == 0) {
    set_last_error("lk: window %d unsupported (odd, <= %d)", win, LK_MAXWIN);
    return PLV_E_CAPACITY;
  }
  ProfScope ps(ctx->prof, "lk_kernel", ctx->stream);
  CamK none{};
  // (measurement: PLV_KNOB_LK_LEGACY_LOOP launches the loop of rounds 2-4 instead, tools/lk_exp.py)
#define LK_LAUNCH(VV)                                                                                                                     \
  hipLaunchKernelGGL(lk_kernel<VV>, dim3(n), dim3(64 * LK4_WAVES), 0, ctx->stream, prev, cur, n, d_pts0, pts1_init ? pts1_init : d_pts1, \
                     d_pts1, d_status, d_iters, win, max_iters, eps, K ? *K : none, K ? d_n0 : nullptr, K ? d_n1 : nullptr)
  if (plv::knob(plv::PLV_KNOB_LK_LEGACY_LOOP))
    LK_LAUNCH(0);
  else if (!plv::knob(plv::PLV_KNOB_LK_AHEAD) || win * win > 64 * LK4_WAVES)
    LK_LAUNCH(1);
  else  // (round 6 experiment, measured no faster: see lk_ahead_kernel) templates of all levels first, search tiles requested a level ahead
    hipLaunchKernelGGL(lk_ahead_kernel, dim3(n), dim3(64 * LK4_WAVES), 0, ctx->stream, prev, cur, n, d_pts0, pts1_init ? pts1_init : d_pts1, d_pts1,
                       d_status, d_iters, win, max_iters, eps, K ? *K : none, K ? d_n0 : nullptr, K ? d_n1 : nullptr);
#undef LK_LAUNCH
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_undistort(plv_ctx *ctx, const CamK &K, int n, const float *d_uv, float *d_xy) {
  ProfScope ps(ctx->prof, "undistort_kernel", ctx->stream);
  hipLaunchKernelGGL(undistort_kernel, dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, K, n, d_uv, d_xy);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}
int launch_undistort2(plv_ctx *ctx, const CamK &K, int n, const float *d_uv0, const float *d_uv1, float *d_xy0,
                      float *d_xy1) {
  ProfScope ps(ctx->prof, "undistort_kernel", ctx->stream);
  hipLaunchKernelGGL(undistort2_kernel, dim3(cdiv(2 * n, 64)), dim3(64), 0, ctx->stream, K, n, d_uv0, d_uv1, d_xy0, d_xy1);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_ransac(plv_ctx *ctx, const float *d_m1, const float *d_m2, int n, double thr, double conf, int max_iters,
                  unsigned seed, int *d_counts, const uint8_t *d_klt, uint8_t *d_mask, int *d_info, double *d_models,
                  const void *mir_src, void *mir_dst, size_t mir_bytes, uint8_t *mir_mask, bool *mirrored, unsigned *done_word, unsigned done_val) {
  if (mirrored) *mirrored = false;
  const float t = (float)(thr * thr);
  if (n < 7) {
    PLV_HIP_CHECK(hipMemsetAsync(d_mask, 0, n, ctx->stream));
    PLV_HIP_CHECK(hipMemsetAsync(d_info, 0, 2 * sizeof(int), ctx->stream));
    return PLV_OK;
  }
  const int nh = n == 7 ? 1 : max_iters;
  {
    ProfScope ps(ctx->prof, "ransac_hyp_kernel", ctx->stream);
    hipLaunchKernelGGL(ransac_hyp_kernel, dim3(nh), dim3(64), 0, ctx->stream, d_m1, d_m2, n, t, seed, d_counts, d_models);
  }
  {
    ProfScope ps(ctx->prof, "ransac_select_kernel", ctx->stream);
    hipLaunchKernelGGL(ransac_select_kernel, dim3(1), dim3(d_models ? 256 : 64), 0, ctx->stream, d_m1, d_m2, n, t, conf, max_iters, seed,
                       d_counts, d_klt, d_mask, d_info, (const double *)d_models, (const unsigned *)mir_src, (unsigned *)mir_dst,
                       (int)(mir_bytes / 4), mir_mask, mir_dst ? done_word : nullptr, done_val);
    if (mirrored) *mirrored = mir_dst != nullptr;
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

}  // namespace plv
