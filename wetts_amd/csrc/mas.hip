// Monotonic alignment search (utils/monotonic_align.py:22-57 `maximum_path_jit`) on gfx950.
//
// The reference runs a numba loop serially over the batch on the host (with a GPU->CPU->GPU
// round trip, monotonic_align.py:13-19).  Here one workgroup owns one utterance: the forward DP
//     value[y,x] += max(value[y-1,x-1], value[y-1,x])      over the band
//     max(0, t_x+y-t_y) <= x < min(t_x, y+1)
// is parallel across x (one lane per column) and sequential over y, with the previous DP row
// held in LDS so a row step costs one LDS round trip + one barrier; the backtrack is a serial
// walk by lane 0.  Arithmetic is kept operation-for-operation identical to the reference
// (f32 add, -1e9 sentinels, in-place table, cells outside the band left at their raw value),
// so the path is bit-exact, including ties (strict `<` in the backtrack).
#include "kernels.h"

namespace wetts {

__global__ __launch_bounds__(1024) void mas_kernel(const float* __restrict__ neg_cent,
                                                   const int32_t* __restrict__ t_ys,
                                                   const int32_t* __restrict__ t_xs, int Ty, int Tx,
                                                   int32_t* __restrict__ path,
                                                   float* __restrict__ values) {
  extern __shared__ float rows[];  // [2][Tx] previous / current DP row
  const int b = blockIdx.x;
  const int t_y = t_ys[b], t_x = t_xs[b];
  const float* nc = neg_cent + (int64_t)b * Ty * Tx;
  float* val = values + (int64_t)b * Ty * Tx;
  int32_t* pth = path + (int64_t)b * Ty * Tx;
  const float max_neg_val = -1e9f;
  if (t_y <= 0 || t_x <= 0 || t_y > Ty || t_x > Tx) return;

  // the reference copies neg_cent (astype) and updates the copy in place
  for (int64_t i = threadIdx.x; i < (int64_t)t_y * Tx; i += blockDim.x) val[i] = nc[i];
  __syncthreads();

  float* prev = rows;
  float* cur = rows + Tx;
  for (int y = 0; y < t_y; ++y) {
    const int x_lo = max(0, t_x + y - t_y);
    const int x_hi = min(t_x, y + 1);
    // cur row starts as the raw values (cells outside the band stay raw)
    for (int x = threadIdx.x; x < t_x; x += blockDim.x) {
      float v = val[(int64_t)y * Tx + x];
      if (x >= x_lo && x < x_hi) {
        float v_cur = (x == y) ? max_neg_val : prev[x];
        float v_prev;
        if (x == 0) v_prev = (y == 0) ? 0.f : max_neg_val;
        else v_prev = prev[x - 1];
        v = v + fmaxf(v_prev, v_cur);
        val[(int64_t)y * Tx + x] = v;
      }
      cur[x] = v;
    }
    __syncthreads();
    float* tmp = prev;
    prev = cur;
    cur = tmp;
  }

  // backtrack (monotonic_align.py:52-57); value[y-1] with y == 0 wraps to the last row in numpy
  if (threadIdx.x == 0) {
    int index = t_x - 1;
    for (int y = t_y - 1; y >= 0; --y) {
      pth[(int64_t)y * Tx + index] = 1;
      if (index != 0) {
        bool step = (index == y);
        if (!step) {
          const int yr = (y - 1 >= 0) ? (y - 1) : (Ty - 1);
          const float* src = (yr < t_y) ? val : nc;  // rows >= t_y were never copied: raw values
          step = src[(int64_t)yr * Tx + index] < src[(int64_t)yr * Tx + index - 1];
        }
        if (step) index -= 1;
      }
    }
  }
}

int32_t k_mas(const float* neg_cent, const int32_t* t_ys, const int32_t* t_xs, int B, int Ty,
              int Tx, int32_t* path, float* values, hipStream_t s) {
  if (B == 0 || Ty == 0 || Tx == 0) return WETTS_OK;
  WETTS_HIP_CHECK(hipMemsetAsync(path, 0, (size_t)B * Ty * Tx * sizeof(int32_t), s));
  int threads = Tx >= 1024 ? 1024 : ((Tx + 63) / 64) * 64;
  size_t lds = (size_t)2 * Tx * sizeof(float);
  WETTS_REQUIRE(lds <= 64 * 1024, "MAS: Tx=%d too large for the LDS row buffers", Tx);
  hipLaunchKernelGGL(mas_kernel, dim3(B), dim3(threads), lds, s, neg_cent, t_ys, t_xs, Ty, Tx, path,
                     values);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
