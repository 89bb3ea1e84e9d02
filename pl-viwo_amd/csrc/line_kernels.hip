// line_kernels.hip — line detection of TrackLSD::perform_detection_monocular (a10) on the device.
//   REF call site: PL-VIWO/src/update/cam/TrackLSD.cpp:194-235 — cv::resize(0.5, INTER_LINEAR),
//   cv::ximgproc::FastLineDetector(20, sqrt 2, 50, 50, 3, no merge)::detect, x2, FilterShortLines(40).
// The detector is an OpenCV-contrib dependency that is not under /root/reference; the kernels follow its
// published algorithm (Lee et al., ICRA 2014) and cv::Canny's (L1 gradient, 3x3 Sobel, fixed-point
// direction test) — the same restatement the oracle makes, see oracle/line_oracle.cpp.
//
//   half_kernel      exact 2x decimation, (a + b + c + d + 2) >> 2                      HBM-bound, elementwise
//   canny_kernel     Sobel + |gx|+|gy| + non-maximum suppression + double threshold      HBM-bound stencil,
//                    on a 16x16 tile with the 2-pixel halo staged in LDS                 one read of the image
//   canny_hyst_kernel  hysteresis flood (only launched when low != high; the reference uses 50/50)
//   fld_walk_kernel  8-neighbour chain walking.  The walk consumes edge pixels in raster order and
//                    every step depends on the previous one: it is the sequential core of the
//                    detector.  One wave; the edge map lives in LDS (90 KB at 376x240) so a step costs
//                    LDS latency, not HBM latency; the raster seed search tests 64 pixels per step.
//   fld_fit_kernel   segment growing (incremental least-squares fit on exact integer sums), clamping,
//                    orientation by side brightness: one lane per chain, chains are independent.
#include "line_kernels.hpp"

#include "fld_fit_core.hpp"
#include "equalize_lut.hpp"

namespace plv {

__global__ void __launch_bounds__(256) half_kernel(const uint8_t *__restrict__ src, int w, int h, uint8_t *__restrict__ dst) {
  const int w2 = w >> 1, h2 = h >> 1;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w2 || y >= h2) return;
  const uint8_t *p = src + (size_t)(2 * y) * w + 2 * x;
  dst[(size_t)y * w2 + x] = (uint8_t)((p[0] + p[1] + p[w] + p[w + 1] + 2) >> 2);
}

#define CN_T 16
// FULL = true: `src` is the full-resolution image (fw wide) and the tile of the half-resolution image the stencil needs is decimated
// on the fly (the arithmetic of half_kernel); the workgroup also writes its 16 x 16 piece of the half-resolution image, which the
// segment fit reads.  One launch for cv::resize + cv::Canny.
// hist (FULL only, nullable): `src` is the RAW image and hist its 256-bin histogram — the workgroup builds cv::equalizeHist's look-up
// table itself and reads the image through it: the launch then need not wait for the pyramid launch that writes the equalised image.
// ------------------------------------------------------------------------------------------ component labels (helpers)
// Union-find with atomicMin (label equivalence, Playne & Hawick's scheme): L[i] = parent of i, a root is its own parent and the
// smallest index of its set.  Works on LDS (inside canny_kernel) and on global memory (ccl_boundary_kernel) alike.
__device__ __forceinline__ int ccl_find(const int *L, int i) {
  for (;;) {
    const int p = __atomic_load_n(L + i, __ATOMIC_RELAXED);
    if (p == i) return i;
    i = p;
  }
}
__device__ __forceinline__ void ccl_union(int *L, int a, int b) {
  for (;;) {
    a = ccl_find(L, a);
    b = ccl_find(L, b);
    if (a == b) return;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }  // a > b: hang a under b unless someone got there first
    const int old = atomicMin(L + a, b);
    if (old == a) return;
    a = old;
  }
}

template <bool FULL>
__global__ void __launch_bounds__(CN_T *CN_T) canny_kernel(const uint8_t *__restrict__ src, int fw, int w, int h, int low, int high,
                                                            uint8_t *__restrict__ map /* 0 weak, 1 none, 2 edge */,
                                                            uint8_t *__restrict__ half_out, const unsigned *__restrict__ hist, int npix_full,
                                                            int *__restrict__ lab_init /* nullable: component labels, first pass */,
                                                            int *__restrict__ cnt_init /* nullable: zeroed per pixel (ccl_roots_kernel counts in it) */) {
  __shared__ int px[CN_T + 4][CN_T + 4];
  __shared__ int mg[CN_T + 2][CN_T + 2];
  __shared__ unsigned cdf[16];
  __shared__ uint8_t lut[256];
  const bool eq = FULL && hist != nullptr;  // (uniform)
  if (eq) equalize_lut_256(hist, npix_full, cdf, lut);
  const int tx = threadIdx.x & (CN_T - 1), ty = threadIdx.x / CN_T;
  const int x0 = blockIdx.x * CN_T, y0 = blockIdx.y * CN_T;
  for (int i = threadIdx.x; i < (CN_T + 4) * (CN_T + 4); i += CN_T * CN_T) {
    const int ly = i / (CN_T + 4), lx = i - ly * (CN_T + 4);
    const int gx = min(max(x0 + lx - 2, 0), w - 1), gy = min(max(y0 + ly - 2, 0), h - 1);  // BORDER_REPLICATE
    if (FULL) {
      const uint8_t *p = src + (size_t)(2 * gy) * fw + 2 * gx;
      px[ly][lx] = eq ? (lut[p[0]] + lut[p[1]] + lut[p[fw]] + lut[p[fw + 1]] + 2) >> 2 : (p[0] + p[1] + p[fw] + p[fw + 1] + 2) >> 2;
    } else {
      px[ly][lx] = src[(size_t)gy * w + gx];
    }
  }
  __syncthreads();
  if (FULL && half_out && x0 + tx < w && y0 + ty < h) half_out[(size_t)(y0 + ty) * w + x0 + tx] = (uint8_t)px[ty + 2][tx + 2];
  for (int i = threadIdx.x; i < (CN_T + 2) * (CN_T + 2); i += CN_T * CN_T) {
    const int ly = i / (CN_T + 2), lx = i - ly * (CN_T + 2);
    const int gx = x0 + lx - 1, gy = y0 + ly - 1;
    int m = 0;
    if (gx >= 0 && gx < w && gy >= 0 && gy < h) {  // magnitudes outside the image are zero
      const int cx = lx + 1, cy = ly + 1;
      const int sx = (px[cy - 1][cx + 1] - px[cy - 1][cx - 1]) + 2 * (px[cy][cx + 1] - px[cy][cx - 1]) + (px[cy + 1][cx + 1] - px[cy + 1][cx - 1]);
      const int sy = (px[cy + 1][cx - 1] - px[cy - 1][cx - 1]) + 2 * (px[cy + 1][cx] - px[cy - 1][cx]) + (px[cy + 1][cx + 1] - px[cy - 1][cx + 1]);
      m = abs(sx) + abs(sy);
    }
    mg[ly][lx] = m;
  }
  __syncthreads();
  const int x = x0 + tx, y = y0 + ty;
  const bool inside = x < w && y < h;
  if (!lab_init && !inside) return;  // (with labels every thread stays for the tile's barriers)
  int mydx, mydy;  // gradient of this thread's own pixel
  {
    const int cx = tx + 2, cy = ty + 2;
    mydx = (px[cy - 1][cx + 1] - px[cy - 1][cx - 1]) + 2 * (px[cy][cx + 1] - px[cy][cx - 1]) + (px[cy + 1][cx + 1] - px[cy + 1][cx - 1]);
    mydy = (px[cy + 1][cx - 1] - px[cy - 1][cx - 1]) + 2 * (px[cy + 1][cx] - px[cy - 1][cx]) + (px[cy + 1][cx + 1] - px[cy - 1][cx + 1]);
  }
  const int m = mg[ty + 1][tx + 1];
  uint8_t out = 1;
  if (m > low) {
    const int ax = abs(mydx), ay = abs(mydy) << 15;
    const int tg22x = ax * 13573;  // tan(22.5 deg) * 2^15
    bool keep;
    if (ay < tg22x)
      keep = m > mg[ty + 1][tx] && m >= mg[ty + 1][tx + 2];
    else {
      const int tg67x = tg22x + (ax << 16);
      if (ay > tg67x)
        keep = m > mg[ty][tx + 1] && m >= mg[ty + 2][tx + 1];
      else {
        const int s = (mydx ^ mydy) < 0 ? -1 : 1;
        keep = m > mg[ty][tx + 1 - s] && m > mg[ty + 2][tx + 1 + s];
      }
    }
    if (keep) out = m > high ? 2 : 0;
  }
  // FastLineDetector clears the top-left 6x6 and the bottom-right 5x5 corner of the edge map
  if ((x < 6 && y < 6) || (x >= w - 5 && y >= h - 5)) out = 1;
  if (inside) map[(size_t)y * w + x] = out;
  if (lab_init) {
    // component labels, first pass: union-find inside the tile, in LDS (W / NW / N / NE neighbours of the same tile); every edge pixel
    // leaves with the global index of its tile-component's first pixel (ccl_boundary_kernel joins the tiles)
    __shared__ int lab[CN_T * CN_T], lcnt[CN_T * CN_T];
    const int me = ty * CN_T + tx;
    const bool edge = inside && out == 2;
    lab[me] = edge ? me : -1;
    lcnt[me] = 0;
    __syncthreads();
    if (edge) {
      if (tx > 0 && lab[me - 1] >= 0) ccl_union(lab, me, me - 1);
      if (ty > 0) {
        if (tx > 0 && lab[me - CN_T - 1] >= 0) ccl_union(lab, me, me - CN_T - 1);
        if (lab[me - CN_T] >= 0) ccl_union(lab, me, me - CN_T);
        if (tx + 1 < CN_T && lab[me - CN_T + 1] >= 0) ccl_union(lab, me, me - CN_T + 1);
      }
    }
    __syncthreads();
    int r = -1;
    if (edge) {
      r = ccl_find(lab, me);
      atomicAdd(&lcnt[r], 1);  // pixels of the tile-component, at its first pixel
    }
    __syncthreads();
    if (inside) {
      lab_init[(size_t)y * w + x] = edge ? (y0 + r / CN_T) * w + x0 + (r & (CN_T - 1)) : -1;
      if (cnt_init) cnt_init[(size_t)y * w + x] = lcnt[me];  // (0 everywhere but at the first pixel of a tile-component)
    }
  }
}

// ------------------------------------------------------------------------------------------ component labels
// 8-connected components of the edge map, so that the host stage of the detector can split its work (line_host.hpp detect_part: the
// chain walk never leaves a component).  canny_kernel has united the pixels of every 16 x 16 tile; ccl_boundary_kernel unites across
// tile borders (global union-find, only the pixels on a tile's first row / first column / last column take part);
// ccl_roots_kernel flattens, counts the pixels of every component and lists the roots; ccl_assign_kernel gives the CCL_BIG largest
// components a part of their own (1 .. CCL_BIG: they bound the host stage's longest thread) and hashes the others into CCL_HASHED
// more; ccl_flatten_kernel writes the part of every pixel (0: not an edge) into the host-visible map.
#define CCL_BIG 8
#define CCL_HASHED 16
#define CCL_ROOT_CAP 4096
#define CCL_PARTS (CCL_BIG + CCL_HASHED)
__global__ void __launch_bounds__(256) ccl_boundary_kernel(int *__restrict__ L, int w, int h, int *__restrict__ n_roots) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *n_roots = 0;
  if (i >= w * h || L[i] < 0) return;
  const int y = i / w, x = i - y * w;
  const int lx = x & (CN_T - 1), ly = y & (CN_T - 1);
  if (lx == 0 && x > 0 && L[i - 1] >= 0) ccl_union(L, i, i - 1);
  if (y > 0) {
    if ((lx == 0 || ly == 0) && x > 0 && L[i - w - 1] >= 0) ccl_union(L, i, i - w - 1);
    if (ly == 0 && L[i - w] >= 0) ccl_union(L, i, i - w);
    if ((ly == 0 || lx == CN_T - 1) && x + 1 < w && L[i - w + 1] >= 0) ccl_union(L, i, i - w + 1);
  }
}
__global__ void __launch_bounds__(256) ccl_roots_kernel(int *__restrict__ L, int n, int *__restrict__ cnt, int *__restrict__ roots, int *__restrict__ n_roots) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || L[i] < 0) return;
  const int r = ccl_find(L, i);
  if (r != i) {
    __atomic_store_n(L + i, r, __ATOMIC_RELAXED);  // (path compression: still an ancestor for whoever reads it meanwhile)
    const int c = cnt[i];  // canny_kernel left the size of a tile-component at its first pixel: one atomic per tile-component
    if (c > 0) {
      atomicAdd(cnt + r, c);
      cnt[i] = 0;  // (nobody adds into a pixel that is not a root)
    }
  }
  if (r == i) {
    const int at = atomicAdd(n_roots, 1);
    if (at < CCL_ROOT_CAP) roots[at] = i;
  }
}
// part of a component: 1 + its rank by size when it is one of the CCL_BIG largest (ties: smaller root first), else
// 1 + CCL_BIG + hash(root) % CCL_HASHED.  One workgroup; written into cnt[root] as -(part) (the counts are no longer needed).
// min_px: a component with fewer pixels cannot hold a chain the detector keeps (a chain's pixels are distinct pixels of ONE component and
// a kept chain has length_threshold + 1 of them or more), and what its short chains consume no other component can reach: such a
// component gets part 255 — no part's — and the host stage never walks it (most components are that small: a fifth of the edge pixels)
__global__ void __launch_bounds__(1024) ccl_assign_kernel(int *__restrict__ cnt, const int *__restrict__ roots, const int *__restrict__ n_roots, int min_px) {
  // only a component of 64 pixels or more can matter for the balance (most have a handful): those are collected into a short list
  // and ranked among themselves; when fewer than CCL_BIG are that large the rest of the own parts stay empty
  __shared__ int lsz[512], lrt[512];
  __shared__ int n_large;
  const int n = min(*n_roots, CCL_ROOT_CAP);
  const bool listed = *n_roots <= CCL_ROOT_CAP;  // (more components than the list holds: every one is hashed, none gets a part of its own)
  if (threadIdx.x == 0) n_large = 0;
  __syncthreads();
  // (1024 threads: ~900 components, two dependent loads each — one round instead of four; round 6: 12 -> 6 us on the line path's head)
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const int root = roots[j], c = cnt[root];
    bool large = false;
    if (listed && c >= 64) {
      const int at = atomicAdd(&n_large, 1);
      if (at < 512) lsz[at] = c, lrt[at] = root, large = true;
    }
    if (!large) cnt[root] = c < min_px ? -255 : -(1 + CCL_BIG + (int)(((unsigned)root * 2654435761u >> 8) % (unsigned)CCL_HASHED));
  }
  __syncthreads();
  const int m = min(n_large, 512);
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    const int mine = lsz[j], root = lrt[j];
    int rank = 0;
    for (int q = 0; q < m && rank < CCL_BIG; ++q) rank += (lsz[q] > mine) || (lsz[q] == mine && lrt[q] < root);
    cnt[root] = -(rank < CCL_BIG ? 1 + rank : 1 + CCL_BIG + (int)(((unsigned)root * 2654435761u >> 8) % (unsigned)CCL_HASHED));
  }
}
// Next to the part of every pixel the launch leaves, per run of 256 pixels in raster order (= one workgroup), the pixels of every part
// in raster order: blk_sorted[256 * b + ..] = the run's pixels (as offsets 0 .. 255 inside the run) grouped by part, blk_bins[b][p] ..
// blk_bins[b][p + 1] the group of part p + 1.  The host stage then forms a part's private map and its seeds from the part's own
// pixels (detect_part) instead of reading the labels of the whole image for every part.  Counting sort in LDS: a pixel's place is
// its part's start + the part's pixels in the waves before + those in the lanes before (ballots).
__global__ void __launch_bounds__(256) ccl_flatten_kernel(const int *__restrict__ L, int n, const int *__restrict__ cnt, uint8_t *__restrict__ lab_out, int min_px,
                                                          uint8_t *__restrict__ blk_sorted, unsigned short *__restrict__ blk_bins) {
  __shared__ unsigned short wcnt[4][CCL_PARTS], start[CCL_PARTS + 1];
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint8_t v = 0;
  if (i < n) {
    const int r = L[i];
    if (r >= 0) {
      const int c = cnt[r];  // -(part) for a listed root; a root beyond the list keeps its positive count
      v = (uint8_t)(c < 0 ? -c : (c < min_px ? 255 : 1 + CCL_BIG + (int)(((unsigned)r * 2654435761u >> 8) % (unsigned)CCL_HASHED)));
    }
    lab_out[i] = v;
  }
  if (!blk_sorted) return;  // (uniform)
  const int part = (v >= 1 && v <= CCL_PARTS) ? v - 1 : -1, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int before = 0;
  for (int p = 0; p < CCL_PARTS; ++p) {
    const unsigned long long m = __ballot(part == p);
    if (part == p) before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wave][p] = (unsigned short)__popcll(m);
  }
  __syncthreads();
  if (wave == 0) {
    int tot = lane < CCL_PARTS ? wcnt[0][lane] + wcnt[1][lane] + wcnt[2][lane] + wcnt[3][lane] : 0, inc = tot;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const int o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane < CCL_PARTS) start[lane + 1] = (unsigned short)inc;
    if (lane == 0) start[0] = 0;
  }
  __syncthreads();
  if (threadIdx.x <= CCL_PARTS) blk_bins[(size_t)blockIdx.x * (CCL_PARTS + 1) + threadIdx.x] = start[threadIdx.x];
  if (part >= 0) {
    int at = start[part] + before;
    for (int w = 0; w < wave; ++w) at += wcnt[w][part];
    blk_sorted[(size_t)blockIdx.x * 256 + at] = (uint8_t)threadIdx.x;
  }
}

// hysteresis: promote weak pixels adjacent to an edge until nothing changes (one workgroup sweep loop)
__global__ void __launch_bounds__(1024) canny_hyst_kernel(uint8_t *__restrict__ map, int w, int h) {
  __shared__ int changed;
  do {
    __syncthreads();
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < w * h; i += blockDim.x) {
      if (map[i] != 0) continue;
      const int y = i / w, x = i - y * w;
      bool near = false;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int qx = x + dx, qy = y + dy;
          if (qx >= 0 && qy >= 0 && qx < w && qy < h) near = near || map[(size_t)qy * w + qx] == 2;
        }
      if (near) {
        map[i] = 2;
        changed = 1;
      }
    }
    __syncthreads();
  } while (changed);
}

// ------------------------------------------------------------------------------------------ chain walking
// FastLineDetectorImpl::lineDetection's seed loop + getPointChain.  All lanes run the same scalar
// walk (uniform control flow, broadcast LDS reads); the lanes only split up the raster seed search.
template <bool IN_LDS>
__global__ void __launch_bounds__(64) fld_walk_kernel(const uint8_t *__restrict__ map, int w, int h, int length_threshold,
                                                      uint8_t *__restrict__ gwork /* w*h scratch when !IN_LDS */,
                                                      int2 *__restrict__ pts, FldChain *__restrict__ chains, int chain_cap,
                                                      int *__restrict__ counts /* [0] chains, [1] segment slots, [2] points */) {
  extern __shared__ uint8_t lmap[];
  const int lane = threadIdx.x;
  const int npix = w * h;
  auto ld = [&](int i) -> int { return IN_LDS ? (int)lmap[i] : (int)gwork[i]; };
  auto st = [&](int i, uint8_t v) {
    if (IN_LDS)
      lmap[i] = v;
    else
      gwork[i] = v;
  };
  for (int i = lane; i < npix; i += 64) st(i, map[i] == 2 ? 255 : 0);
  __syncthreads();
  int n_chain = 0, n_slot = 0, n_pts = 0;
  const int dxs[8] = {1, 0, -1, -1, -1, 0, 1, 1}, dys[8] = {1, 1, 1, 0, -1, -1, -1, 0};  // {row,col} pairs of the reference
  int scan = 0;
  while (scan < npix) {
    // raster search for the next seed: 64 pixels at a time
    const int idx = scan + lane;
    const unsigned long long m = __ballot(idx < npix && ld(min(idx, npix - 1)) != 0);
    if (m == 0) {
      scan += 64;
      continue;
    }
    const int seed = scan + __ffsll((long long)m) - 1;
    int x = seed % w, y = seed / w;
    const int start = n_pts;
    if (lane == 0) {
      pts[n_pts] = make_int2(x, y);
      st(seed, 0);
    }
    ++n_pts;
    __syncthreads();
    float direction = 0.0f;
    int step = 0;
    for (;;) {
      // the eight neighbours first (clamped addresses, all loads in flight together), decisions afterwards
      bool valid[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ci = x + dxs[i], ri = y + dys[i];
        const bool inb = ri >= 0 && ri < h && ci >= 0 && ci < w;
        const int v = ld(min(max(ri, 0), h - 1) * w + min(max(ci, 0), w - 1));
        valid[i] = inb && v != 0;
      }
      int pick = -1;
      if (step == 0) {
#pragma unroll
        for (int i = 7; i >= 0; --i) pick = valid[i] ? i : pick;  // first valid neighbour
        if (pick < 0) break;
        direction = pick > 4 ? (float)(pick - 8) : (float)pick;
      } else {
        float min_dir_diff = 7.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float curr = i > 4 ? (float)(i - 8) : (float)i;
          float diff = fabsf(curr - direction);
          diff = diff > 4.0f ? 8.0f - diff : diff;
          const bool take = valid[i] && diff <= min_dir_diff;  // ties: the later neighbour wins
          min_dir_diff = take ? diff : min_dir_diff;
          pick = take ? i : pick;
        }
        if (pick < 0 || !(min_dir_diff < 2.0f)) break;
        const int cdir = pick > 4 ? pick - 8 : pick;
        direction = (direction * (float)step + (float)cdir) / (float)(step + 1);
      }
      x += dxs[pick];
      y += dys[pick];
      if (lane == 0) {
        pts[n_pts] = make_int2(x, y);
        st(y * w + x, 0);
      }
      ++n_pts;
      ++step;
      __syncthreads();
    }
    const int len = n_pts - start;
    if (len >= length_threshold + 1 && n_chain < chain_cap) {
      if (lane == 0) chains[n_chain] = FldChain{start, len, n_slot};
      n_slot += len / length_threshold + 1;
      ++n_chain;
    } else {
      n_pts = start;  // too short: forget the points
    }
    scan = seed + 1;
  }
  if (lane == 0) {
    counts[0] = n_chain;
    counts[1] = n_slot;
    counts[2] = n_pts;
  }
}

// ------------------------------------------------------------------------------------------ segment growing
__global__ void __launch_bounds__(64) fld_fit_kernel(const uint8_t *__restrict__ src, int w, int h, int length_threshold,
                                                     float distance_threshold, const int2 *__restrict__ pts,
                                                     const FldChain *__restrict__ chains, const int *__restrict__ counts,
                                                     float4 *__restrict__ segs, int *__restrict__ seg_count) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= counts[0]) return;
  const FldChain ch = chains[c];
  seg_count[c] = fit_chain(src, w, h, length_threshold, distance_threshold, pts + ch.start, ch.len, segs + ch.slot);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// stage 1: half-resolution image and Canny map (0 weak / 1 none / 2 edge; hysteresis applied)
int launch_line_edges(plv_ctx *ctx, const uint8_t *d_img, int W, int H, const FldParams &fp, FldBuffers &b, hipStream_t st, const unsigned *d_hist) {
  if (!st) st = ctx->stream;
  const int w = W / 2, h = H / 2;
  int low = fp.canny_low, high = fp.canny_high;
  if (low > high) std::swap(low, high);
  {
    ProfScope ps(ctx->prof, "half_canny_kernel", st);
    hipLaunchKernelGGL(canny_kernel<true>, dim3(cdiv(w, CN_T), cdiv(h, CN_T)), dim3(CN_T * CN_T), 0, st, d_img, W, w, h, low,
                       high, b.map, b.half, d_hist, W * H, b.lab_work, b.lab_cnt);
  }
  if (low != high) {
    ProfScope ps(ctx->prof, "canny_hyst_kernel", st);
    hipLaunchKernelGGL(canny_hyst_kernel, dim3(1), dim3(1024), 0, st, b.map, w, h);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

// stage 1b: component labels of the edge map (canny_kernel left the tile-local pass in b.lab_work) as parts 1 .. kLineParts per
// edge pixel in b.lab_out (host-visible), on stream st
int launch_line_labels(plv_ctx *ctx, int w, int h, FldBuffers &b, hipStream_t st, int length_threshold) {
  const int n = w * h;
  const int min_px = length_threshold + 1;  // (fld_walk: a chain is kept when it has length_threshold + 1 points or more)
  {
    ProfScope ps(ctx->prof, "ccl_boundary_kernel", st);
    hipLaunchKernelGGL(ccl_boundary_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, b.lab_work, w, h, b.lab_roots + CCL_ROOT_CAP);
  }
  {
    ProfScope ps(ctx->prof, "ccl_roots_kernel", st);
    hipLaunchKernelGGL(ccl_roots_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, b.lab_work, n, b.lab_cnt, b.lab_roots, b.lab_roots + CCL_ROOT_CAP);
  }
  {
    ProfScope ps(ctx->prof, "ccl_assign_kernel", st);
    hipLaunchKernelGGL(ccl_assign_kernel, dim3(1), dim3(1024), 0, st, b.lab_cnt, b.lab_roots, b.lab_roots + CCL_ROOT_CAP, min_px);
  }
  {
    ProfScope ps(ctx->prof, "ccl_flatten_kernel", st);
    hipLaunchKernelGGL(ccl_flatten_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, b.lab_work, n, b.lab_cnt, b.lab_out, min_px, b.blk_sorted, b.blk_bins);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}
int line_label_parts() { return CCL_PARTS; }
size_t line_label_roots_bytes() { return (CCL_ROOT_CAP + 4) * sizeof(int); }

// stage 2 (device variant): chain walking by one wave
int launch_line_walk(plv_ctx *ctx, int w, int h, const FldParams &fp, FldBuffers &b) {
  ProfScope ps(ctx->prof, "fld_walk_kernel", ctx->stream);
  const size_t need = (size_t)w * h;
  if (need <= 150 * 1024) {
    PLV_HIP_CHECK(ensure_dyn_smem((const void *)fld_walk_kernel<true>, (int)need));
    hipLaunchKernelGGL(fld_walk_kernel<true>, dim3(1), dim3(64), need, ctx->stream, b.map, w, h, fp.length_threshold, b.work, b.pts,
                       b.chains, b.chain_cap, b.counts);
  } else {
    hipLaunchKernelGGL(fld_walk_kernel<false>, dim3(1), dim3(64), 0, ctx->stream, b.map, w, h, fp.length_threshold, b.work, b.pts,
                       b.chains, b.chain_cap, b.counts);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

// stage 3: one lane per chain
int launch_line_fit(plv_ctx *ctx, int w, int h, const FldParams &fp, FldBuffers &b) {
  ProfScope ps(ctx->prof, "fld_fit_kernel", ctx->stream);
  hipLaunchKernelGGL(fld_fit_kernel, dim3(cdiv(b.chain_cap, 64)), dim3(64), 0, ctx->stream, b.half, w, h, fp.length_threshold,
                     fp.distance_threshold, b.pts, b.chains, b.counts, b.segs, b.seg_count);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

}  // namespace plv
