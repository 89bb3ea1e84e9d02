// detect_kernels.hip — K4 (FAST-9/16 + 3x3 NMS + per-cell top-k) and K5 (cornerSubPix).
//
//   fast_cells_kernel  cv::FAST(img(roi), thr, nms=true) + sort + top-k per grid cell
//                      REF: open_vins/ov_core/src/track/Grider_GRID.h:108-151
//   subpix_kernel      cv::cornerSubPix(5x5 window, (-1,-1), 20 it | 1e-3)   REF: Grider_GRID.h:163-174
//
// FAST runs on every requested cell's ROI exactly like the reference (3-px border of the ROI skipped, so corners next
// to a cell edge are never found — reproduced), as 32 x 32 tiles (fast_tiles_kernel); the per-cell "sort by response, keep
// the first num_features_grid" is a rank count on the key (score, raster order) per cell (fast_topk_kernel), which also fixes
// the tie order that std::sort leaves unspecified.
#include "detect_kernels.hpp"
#include "wave_ops.hpp"

namespace plv {

__constant__ int FAST_OFF[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// largest threshold that keeps (x,y) a FAST-9/16 corner, minus 1; 0 if not a corner at `thr`
__device__ __forceinline__ int fast_score_lds(const uint8_t *roi, int pitch, int x, int y, int thr) {
  int d[16];
  const int v = roi[y * pitch + x];
#pragma unroll
  for (int k = 0; k < 16; ++k) d[k] = v - roi[(y + FAST_OFF[k][1]) * pitch + x + FAST_OFF[k][0]];
  // quick reject (Rosten): a 9-arc must contain pixel 0 or 8, 4 or 12 ... on the same side
  int best_dark = -1000, best_bright = -1000;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    int mn = d[k], mx = d[k];
#pragma unroll
    for (int q = 1; q < 9; ++q) {
      const int e = d[(k + q) & 15];
      mn = min(mn, e);
      mx = max(mx, e);
    }
    best_dark = max(best_dark, mn);
    best_bright = max(best_bright, -mx);
  }
  const int a = max(best_dark, best_bright);
  return a <= thr ? 0 : a - 1;
}

// ---- K4, stage 1: scores + non-maximum suppression on 32 x 32 tiles of every requested cell (n_cells * tiles workgroups: the
// chip is filled, where one workgroup per cell used 25 CUs).  The tile and a 4-pixel ring are staged in LDS (3 for the FAST circle,
// 1 for the neighbours of the suppression); a cheap necessary test (a 9-arc of 16 contains one pixel of every antipodal pair) sends
// ~5 % of the pixels to the full score through a compacted list, so that the lanes of a wave stay busy; local maxima go to the
// cell's candidate list as keys (score << 32 | ~raster index): a larger key comes earlier in the reference's sorted order
// (response descending, raster ascending) and keys are unique, so the order they arrive in does not matter.
#define FAST_TILE 32
__global__ void __launch_bounds__(256) fast_tiles_kernel(DetectParams P, int tiles_x, int tiles_y, unsigned long long *__restrict__ cand,
                                                         int *__restrict__ cand_n) {
  constexpr int RW = FAST_TILE + 8, SW = FAST_TILE + 2;
  __shared__ uint8_t roi[RW * RW];
  __shared__ short sc[SW * SW];
  __shared__ short list[SW * SW];
  __shared__ int nlist;
  const int per_cell = tiles_x * tiles_y;
  const int cell = blockIdx.x / per_cell, tile = blockIdx.x - cell * per_cell;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int cw = P.cell_w, chh = P.cell_h;
  const int x0 = P.cells[2 * cell] * cw, y0 = P.cells[2 * cell + 1] * chh;
  const int lx0 = tx * FAST_TILE, ly0 = ty * FAST_TILE;
  const int t = threadIdx.x;
  if (t == 0) nlist = 0;
  for (int i = t; i < RW * RW; i += 256) {
    const int ry = i / RW, rx = i - ry * RW;
    const int lx = min(max(lx0 + rx - 4, 0), cw - 1), ly = min(max(ly0 + ry - 4, 0), chh - 1);
    roi[i] = P.img[(size_t)(y0 + ly) * P.W + x0 + lx];
  }
  for (int i = t; i < SW * SW; i += 256) sc[i] = 0;
  __syncthreads();
  const int thr = P.threshold;
  for (int i = t; i < SW * SW; i += 256) {
    const int sy = i / SW, sx = i - sy * SW;
    const int lx = lx0 + sx - 1, ly = ly0 + sy - 1;
    if (lx < 3 || lx >= cw - 3 || ly < 3 || ly >= chh - 3) continue;  // FAST skips a 3-pixel border of the ROI it is given
    const uint8_t *c = &roi[(sy + 3) * RW + sx + 3];
    const int v = c[0];
    const int d0 = v - c[3 * RW], d8 = v - c[-3 * RW], d4 = v - c[3], d12 = v - c[-3];
    const bool dark = (d0 > thr || d8 > thr) && (d4 > thr || d12 > thr);
    const bool bright = (-d0 > thr || -d8 > thr) && (-d4 > thr || -d12 > thr);
    if (dark || bright) list[atomicAdd(&nlist, 1)] = (short)i;
  }
  __syncthreads();
  const int nl = nlist;
  for (int j = t; j < nl; j += 256) {
    const int i = list[j];
    const int sy = i / SW, sx = i - sy * SW;
    sc[i] = (short)fast_score_lds(roi, RW, sx + 3, sy + 3, thr);
  }
  __syncthreads();
  for (int i = t; i < FAST_TILE * FAST_TILE; i += 256) {
    const int yy = i / FAST_TILE, xx = i - yy * FAST_TILE;
    const short *q = &sc[(yy + 1) * SW + xx + 1];
    const int s = q[0];
    if (s == 0) continue;
    // strict 3x3 non-max suppression
    if (q[-SW - 1] >= s || q[-SW] >= s || q[-SW + 1] >= s || q[-1] >= s || q[1] >= s || q[SW - 1] >= s || q[SW] >= s || q[SW + 1] >= s) continue;
    const int lx = lx0 + xx, ly = ly0 + yy;
    const int slot = atomicAdd(&cand_n[cell], 1);
    if (slot < P.cand_cap)
      cand[(size_t)cell * P.cand_cap + slot] = ((unsigned long long)(unsigned)s << 32) | (unsigned)(0x7fffffff - (ly * cw + lx));
  }
}

// ---- K4, stage 2: the reference's "sort by response, keep the first num_features_grid" per cell.  Every candidate counts the
// keys above its own: that is its position in the sorted order, the first nfg write their slot.  Then the bounds / mask /
// occupied-box tests of Grider_GRID.h:141-147 and TrackKLT.cpp:455-461 per kept corner.
__global__ void __launch_bounds__(256) fast_topk_kernel(DetectParams P, const unsigned long long *__restrict__ cand, int *__restrict__ cand_n) {
  // Round 4: the rank of a key is only wanted when it is below nfg (REF Grider_GRID.h:125-128 keeps the first num_features_grid of the
  // sorted cell), and a key's rank is the number of LARGER keys — so only keys at or above the score of the nfg-th best can have one,
  // and only such keys count towards it.  A 256-bin histogram of the scores (a FAST score is at most 255) gives that cut-off score; the
  // rank count then runs over the survivors alone: ~nfg + ties keys instead of the cell's ~2000 local maxima.  Same ranks, same
  // output slots; a cell whose maxima all share one score costs what it used to.
  extern __shared__ unsigned long long keys[];
  __shared__ int hist[256], wtot[4], wcut[4], n_sel;
  const int cell = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nc = min(cand_n[cell], P.cand_cap);
  int *sel = (int *)(keys + P.cand_cap);  // indices of the survivors
  const int cw = P.cell_w, chh = P.cell_h;
  const int x0 = P.cells[2 * cell] * cw, y0 = P.cells[2 * cell + 1] * chh;
  hist[t] = 0;
  if (t == 0) n_sel = 0;
  for (int r = nc + t; r < P.nfg; r += 256) P.out_valid[cell * P.nfg + r] = 0;
  __syncthreads();
  for (int i = t; i < nc; i += 256) {
    const unsigned long long k = cand[(size_t)cell * P.cand_cap + i];
    keys[i] = k;
    atomicAdd(&hist[min((int)(k >> 32), 255)], 1);
  }
  __syncthreads();
  if (t == 0) cand_n[cell] = 0;  // ready for the next frame (stream order: stage 1 of the next detection comes after this kernel)
  // suffix counts S[t] = keys with score >= t (scores above 255 sit in bin 255), then the largest t with S[t] >= nfg
  int v = hist[t];
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_down(v, off);
    if (lane + off < 64) v += u;
  }
  if (lane == 0) wtot[wave] = v;
  __syncthreads();
  for (int w = wave + 1; w < 4; ++w) v += wtot[w];
  const unsigned long long ok_mask = __ballot(v >= P.nfg);
  if (lane == 0) wcut[wave] = ok_mask ? 64 * wave + 63 - __clzll((long long)ok_mask) : -1;
  __syncthreads();
  const int cut = max(max(wcut[0], wcut[1]), max(wcut[2], wcut[3]));  // -1: fewer than nfg keys in the cell, all of them rank
  for (int i = t; i < nc; i += 256)
    if (min((int)(keys[i] >> 32), 255) >= cut) sel[atomicAdd(&n_sel, 1)] = i;
  __syncthreads();
  const int M = n_sel;
  // ranks of the survivors; the keys that get an output slot (rank < nfg) are collected
  unsigned long long *wkey = (unsigned long long *)(sel + P.cand_cap + (P.cand_cap & 1));  // [nfg] (8-byte aligned: the index list is padded to even)
  int *wrank = (int *)(wkey + P.nfg);                                                       // [nfg]
  __shared__ int n_win;
  if (t == 0) n_win = 0;
  __syncthreads();
  for (int q = t; q < M; q += 256) {
    const unsigned long long k = keys[sel[q]];
    int rank = 0;
    for (int j = 0; j < M; ++j) rank += keys[sel[j]] > k;
    if (rank >= P.nfg) continue;
    const int w = atomicAdd(&n_win, 1);  // (< nfg: ranks below nfg are distinct)
    wkey[w] = k, wrank[w] = rank;
  }
  __syncthreads();
  // REF TrackKLT.cpp:486-520: a detection inside the mask or within min_px_dist of an existing point is dropped — the test against the
  // existing points by the whole workgroup per winner (it used to be one thread walking all of them per winner: 500 dependent loads at
  // configs[3], the launch's slowest thread and most of its 90 us)
  const int nw = n_win;
  for (int w = 0; w < nw; ++w) {
    const unsigned long long k = wkey[w];
    const int idx = 0x7fffffff - (int)(k & 0xffffffffULL);
    const int y = idx / cw, x = idx - y * cw;
    const float gx = (float)x + (float)x0, gy = (float)y + (float)y0;
    bool ok = !((int)gx < 0 || (int)gx > P.W || (int)gy < 0 || (int)gy > P.H);
    const int ix = (int)gx, iy = (int)gy;
    if (ok && P.mask && P.mask[(size_t)iy * P.W + ix] > 127) ok = false;
    int near = 0;
    if (ok)  // (uniform)
      for (int b = t; b < P.n_boxes; b += 256) near |= (abs(ix - P.boxes[2 * b]) <= P.min_px_dist && abs(iy - P.boxes[2 * b + 1]) <= P.min_px_dist) ? 1 : 0;
    near = __syncthreads_or(near);
    if (t == 0) {
      const int slot = cell * P.nfg + wrank[w];
      P.out_xy[2 * slot] = gx;
      P.out_xy[2 * slot + 1] = gy;
      P.out_resp[slot] = (float)(unsigned)(k >> 32);
      P.out_valid[slot] = (ok && !near) ? 1 : 0;
    }
  }
}

// One wavefront per candidate slot.  121 window pixels over 64 lanes, double-precision DPP sums.
__global__ void __launch_bounds__(64) subpix_kernel(const uint8_t *__restrict__ img, int W, int H, int n,
                                                    const uint8_t *__restrict__ valid, float *__restrict__ xy,
                                                    const float *__restrict__ mask, int win, int max_iters, double eps) {
  __shared__ float sub[13 * 13];
  __shared__ double term[5][121], tot[5];
  const int s = blockIdx.x;
  if (s >= n || !valid[s]) return;
  const int lane = threadIdx.x;
  const int ww = 2 * win + 1, sw = ww + 2;
  const float cTx = xy[2 * s], cTy = xy[2 * s + 1];
  float cIx = cTx, cIy = cTy;
  const double eps2 = eps * eps;
  int iter = 0;
  double err = 0;
  do {
    const float cx = cIx - (sw - 1) * 0.5f, cy = cIy - (sw - 1) * 0.5f;
    const int ix = (int)floorf(cx), iy = (int)floorf(cy);
    const float a = cx - ix, b = cy - iy;
    const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    __syncthreads();
    for (int i = lane; i < sw * sw; i += 64) {
      const int r = i / sw, c = i - r * sw;
      const int xa = min(max(ix + c, 0), W - 1), xb = min(max(ix + c + 1, 0), W - 1);
      const int ya = min(max(iy + r, 0), H - 1), yb = min(max(iy + r + 1, 0), H - 1);
      sub[i] = (float)img[(size_t)ya * W + xa] * a11 + (float)img[(size_t)ya * W + xb] * a12 + (float)img[(size_t)yb * W + xa] * a21 +
               (float)img[(size_t)yb * W + xb] * a22;
    }
    __syncthreads();
    // the five sums of the normal equations, each added up in raster order by one lane: the result is the serial loop's, bit for bit
    for (int i = lane; i < ww * ww; i += 64) {
      const int r = i / ww, c = i - r * ww;
      const float m = mask[i];
      const float *sp = &sub[(r + 1) * sw + c + 1];
      const float tgx = sp[1] - sp[-1], tgy = sp[sw] - sp[-sw];
      const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
      const double pxx = c - win, pyy = r - win;
      term[0][i] = gxx;
      term[1][i] = gxy;
      term[2][i] = gyy;
      term[3][i] = gxx * pxx + gxy * pyy;
      term[4][i] = gxy * pxx + gyy * pyy;
    }
    __syncthreads();
    if (lane < 5) {
      double acc = 0;
      for (int i0 = 0; i0 < 121; i0 += 11) {  // (win = 5: 11 rows of 11; a row's loads are issued together, the adds stay in raster order)
        double v[11];
#pragma unroll
        for (int u = 0; u < 11; ++u) v[u] = term[lane][i0 + u];
#pragma unroll
        for (int u = 0; u < 11; ++u) acc += v[u];
      }
      tot[lane] = acc;
    }
    __syncthreads();
    const double A = tot[0], B = tot[1], Cc = tot[2], bb1 = tot[3], bb2 = tot[4];
    const double det = A * Cc - B * B;
    if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
    const double scale = 1.0 / det;
    const float nx = (float)(cIx + Cc * scale * bb1 - B * scale * bb2);
    const float ny = (float)(cIy - B * scale * bb1 + A * scale * bb2);
    err = (double)(nx - cIx) * (nx - cIx) + (double)(ny - cIy) * (ny - cIy);
    cIx = nx;
    cIy = ny;
    if (cIx < 0 || cIx >= W || cIy < 0 || cIy >= H) break;
  } while (++iter < max_iters && err > eps2);
  if (fabsf(cIx - cTx) > win || fabsf(cIy - cTy) > win) {
    cIx = cTx;
    cIy = cTy;
  }
  if (lane == 0) {
    xy[2 * s] = cIx;
    xy[2 * s + 1] = cIy;
  }
}

int launch_fast_cells(plv_ctx *ctx, const DetectParams &P, int n_cells, unsigned long long *d_cand, int *d_cand_n) {
  const size_t topk_lds = (size_t)P.cand_cap * 8 + (size_t)(P.cand_cap + (P.cand_cap & 1)) * 4 + (size_t)std::max(P.nfg, 1) * 12;
  if (P.cell_w * P.cell_h > 0x7fffffff / 2 || topk_lds > 60 * 1024) {
    set_last_error("FAST: cell %dx%d / candidate capacity %d out of range", P.cell_w, P.cell_h, P.cand_cap);
    return PLV_E_CAPACITY;
  }
  const int tiles_x = (P.cell_w + FAST_TILE - 1) / FAST_TILE, tiles_y = (P.cell_h + FAST_TILE - 1) / FAST_TILE;
  {
    ProfScope ps(ctx->prof, "fast_tiles_kernel", ctx->stream);
    hipLaunchKernelGGL(fast_tiles_kernel, dim3(n_cells * tiles_x * tiles_y), dim3(256), 0, ctx->stream, P, tiles_x, tiles_y, d_cand, d_cand_n);
  }
  {
    ProfScope ps(ctx->prof, "fast_topk_kernel", ctx->stream);
    hipLaunchKernelGGL(fast_topk_kernel, dim3(n_cells), dim3(256), topk_lds, ctx->stream, P, d_cand, d_cand_n);  // keys + survivor indices + winners
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_subpix(plv_ctx *ctx, const uint8_t *d_img, int W, int H, int n, const uint8_t *d_valid, float *d_xy,
                  const float *d_mask, int win, int max_iters, double eps) {
  if (win != 5) {
    set_last_error("subpix: only the reference's 5x5 half-window is built");
    return PLV_E_CAPACITY;
  }
  ProfScope ps(ctx->prof, "subpix_kernel", ctx->stream);
  hipLaunchKernelGGL(subpix_kernel, dim3(n), dim3(64), 0, ctx->stream, d_img, W, H, n, d_valid, d_xy, d_mask, win, max_iters, eps);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

}  // namespace plv
