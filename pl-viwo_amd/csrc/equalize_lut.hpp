// equalize_lut.hpp — the look-up table of cv::equalizeHist from a 256-bin histogram, by one workgroup of 256 threads (thread = bin).
// REF: OpenCV imgproc/src/histogram.cpp equalizeHist (as called by TrackKLT::feed_monocular, open_vins/ov_core/src/track/TrackKLT.cpp:59):
// lut[i] = saturate(round((cdf[i] - h[i0]) * 255 / (npix - h[i0]))) with i0 the first non-empty bin; the identity when one bin holds
// every pixel.  The prefix sum runs as wave scans: two barriers.  Used by the first pyramid launch (which writes the equalised level 0)
// and by the line detector's edge kernel when it runs ahead of that launch on the raw image (line_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace plv {

// cdf: >= 9 words of LDS scratch, lut: 256 bytes of LDS.  All 256 threads call; ends with a barrier (lut complete).
__device__ __forceinline__ void equalize_lut_256(const unsigned *__restrict__ hist, int npix, unsigned *cdf, uint8_t *lut) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const unsigned hv = hist[t];
  unsigned c = hv;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned u = __shfl_up(c, off);
    if (lane >= off) c += u;
  }
  const unsigned long long nz = __ballot(hv != 0);
  if (lane == 63) cdf[wv] = c;                                                                  // wave totals
  if (lane == 0) cdf[4 + wv] = nz ? (unsigned)(64 * wv + __ffsll((long long)nz) - 1) : 256u;  // first non-empty bin of the wave
  __syncthreads();
  for (int w = 0; w < wv; ++w) c += cdf[w];
  const int i0 = (int)min(min(cdf[4], cdf[5]), min(cdf[6], cdf[7]));
  if (t == i0) cdf[8] = hv;
  __syncthreads();
  const unsigned hh0 = cdf[8];
  if ((int)hh0 == npix) {
    lut[t] = (uint8_t)t;
  } else {
    const float scale = (256 - 1.f) / (float)(npix - (int)hh0);
    int v = 0;
    if (t > i0) v = __float2int_rn((float)(int)(c - hh0) * scale);
    lut[t] = (uint8_t)min(max(v, 0), 255);
  }
  __syncthreads();
}

}  // namespace plv
