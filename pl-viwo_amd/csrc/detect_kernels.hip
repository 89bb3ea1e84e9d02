// detect_kernels.hip — K4 (FAST-9/16 + 3x3 NMS + per-cell top-k) and K5 (cornerSubPix).
//
//   fast_cells_kernel  cv::FAST(img(roi), thr, nms=true) + sort + top-k per grid cell
//                      REF: open_vins/ov_core/src/track/Grider_GRID.h:108-151
//   subpix_kernel      cv::cornerSubPix(5x5 window, (-1,-1), 20 it | 1e-3)   REF: Grider_GRID.h:163-174
//
// One workgroup per grid cell: the cell ROI (e.g. 150 x 96 u8) and its score plane live in LDS,
// FAST runs on the ROI exactly like the reference (3-px border of the ROI skipped, so corners next
// to a cell edge are never found — reproduced).  The per-cell "sort by response, keep the first
// num_features_grid" becomes k rounds of a workgroup arg-max on the key (score, raster order),
// which also fixes the tie order that std::sort leaves unspecified.
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

__global__ void __launch_bounds__(256) fast_cells_kernel(DetectParams P) {
  extern __shared__ unsigned char smem_u8[];
  const int cell = blockIdx.x;
  const int cw = P.cell_w, chh = P.cell_h;
  const int pitch = (cw + 3) & ~3;
  uint8_t *roi = smem_u8;                                           // [chh][pitch]
  short *sc = reinterpret_cast<short *>(smem_u8 + ((chh * pitch + 15) & ~15));  // [chh][cw]
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(sc) + (((size_t)chh * cw * 2 + 15) & ~(size_t)15));
  __shared__ int ncand;
  __shared__ unsigned long long red[4];
  const int x0 = P.cells[2 * cell] * cw, y0 = P.cells[2 * cell + 1] * chh;
  const int t = threadIdx.x;
  if (t == 0) ncand = 0;
  for (int i = t; i < chh * cw; i += 256) {
    const int y = i / cw, x = i - y * cw;
    roi[y * pitch + x] = P.img[(size_t)(y0 + y) * P.W + x0 + x];
  }
  __syncthreads();
  for (int i = t; i < chh * cw; i += 256) {
    const int y = i / cw, x = i - y * cw;
    int s = 0;
    if (x >= 3 && x < cw - 3 && y >= 3 && y < chh - 3) s = fast_score_lds(roi, pitch, x, y, P.threshold);
    sc[i] = (short)s;
  }
  __syncthreads();
  // strict 3x3 non-max suppression -> candidate keys (score << 32 | ~raster index): larger key = earlier in the
  // reference's sorted order (response desc, raster asc)
  for (int i = t; i < chh * cw; i += 256) {
    const int y = i / cw, x = i - y * cw;
    const int s = sc[i];
    if (s == 0) continue;
    bool mx = true;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        if ((dx || dy) && sc[(y + dy) * cw + x + dx] >= s) mx = false;
    if (mx) {
      const int slot = atomicAdd(&ncand, 1);
      if (slot < P.cand_cap) cand[slot] = ((unsigned long long)(unsigned)s << 32) | (unsigned)(0x7fffffff - i);
    }
  }
  __syncthreads();
  const int nc = min(ncand, P.cand_cap);
  // k rounds of arg-max
  for (int r = 0; r < P.nfg; ++r) {
    unsigned long long best = 0;
    for (int i = t; i < nc; i += 256) best = max(best, cand[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffULL), off, 64), hi = __shfl_xor((unsigned)(best >> 32), off, 64);
      best = max(best, ((unsigned long long)hi << 32) | lo);
    }
    if ((t & 63) == 0) red[t >> 6] = best;
    __syncthreads();
    best = max(max(red[0], red[1]), max(red[2], red[3]));
    const int slot = cell * P.nfg + r;
    if (best == 0) {
      if (t == 0) P.out_valid[slot] = 0;
    } else {
      for (int i = t; i < nc; i += 256)
        if (cand[i] == best) cand[i] = 0;  // remove the winner (keys are unique)
      if (t == 0) {
        const int idx = 0x7fffffff - (int)(best & 0xffffffffULL);
        const int y = idx / cw, x = idx - y * cw;
        const float gx = (float)x + (float)x0, gy = (float)y + (float)y0;
        // REF: Grider_GRID.h:141-147 bounds + mask (the mask already carries the +-min_px_dist boxes
        // painted around the tracked points, TrackKLT.cpp:455-461: tested here against the point list)
        bool ok = !((int)gx < 0 || (int)gx > P.W || (int)gy < 0 || (int)gy > P.H);
        const int ix = (int)gx, iy = (int)gy;
        if (ok && P.mask && P.mask[(size_t)iy * P.W + ix] > 127) ok = false;
        for (int q = 0; ok && q < P.n_boxes; ++q)
          if (abs(ix - P.boxes[2 * q]) <= P.min_px_dist && abs(iy - P.boxes[2 * q + 1]) <= P.min_px_dist) ok = false;
        P.out_xy[2 * slot] = gx;
        P.out_xy[2 * slot + 1] = gy;
        P.out_resp[slot] = (float)(unsigned)(best >> 32);
        P.out_valid[slot] = ok ? 1 : 0;
      }
    }
    __syncthreads();
  }
}

// One wavefront per candidate slot.  121 window pixels over 64 lanes, double-precision DPP sums.
__global__ void __launch_bounds__(64) subpix_kernel(const uint8_t *__restrict__ img, int W, int H, int n,
                                                    const uint8_t *__restrict__ valid, float *__restrict__ xy,
                                                    const float *__restrict__ mask, int win, int max_iters, double eps) {
  __shared__ float sub[13 * 13];
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
    double A = 0, B = 0, Cc = 0, bb1 = 0, bb2 = 0;
    for (int i = lane; i < ww * ww; i += 64) {
      const int r = i / ww, c = i - r * ww;
      const float m = mask[i];
      const float *sp = &sub[(r + 1) * sw + c + 1];
      const float tgx = sp[1] - sp[-1], tgy = sp[sw] - sp[-sw];
      const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
      const double pxx = c - win, pyy = r - win;
      A += gxx;
      B += gxy;
      Cc += gyy;
      bb1 += gxx * pxx + gxy * pyy;
      bb2 += gxy * pxx + gyy * pyy;
    }
    A = wave_sum_f64(A);
    B = wave_sum_f64(B);
    Cc = wave_sum_f64(Cc);
    bb1 = wave_sum_f64(bb1);
    bb2 = wave_sum_f64(bb2);
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

int launch_fast_cells(plv_ctx *ctx, const DetectParams &P, int n_cells) {
  const int pitch = (P.cell_w + 3) & ~3;
  size_t shm = ((size_t)P.cell_h * pitch + 15) & ~(size_t)15;
  shm += ((size_t)P.cell_h * P.cell_w * 2 + 15) & ~(size_t)15;
  shm += (size_t)P.cand_cap * 8;
  if (shm > 150 * 1024) {
    set_last_error("FAST: cell %dx%d does not fit LDS", P.cell_w, P.cell_h);
    return PLV_E_CAPACITY;
  }
  PLV_HIP_CHECK(ensure_dyn_smem((const void *)fast_cells_kernel, (int)shm));
  ProfScope ps(ctx->prof, "fast_cells_kernel", ctx->stream);
  hipLaunchKernelGGL(fast_cells_kernel, dim3(n_cells), dim3(256), shm, ctx->stream, P);
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
