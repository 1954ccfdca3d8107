/* ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Plain-C restatement of the reference's numba kernel `maximum_path_jit`
 * (wenet-e2e/wetts, wetts/vits/utils/monotonic_align.py:22-57): in-place Viterbi over the band
 * max(0, t_x+y-t_y) <= x < min(t_x, y+1), then the backtrack with a strict `<`.
 * Pinned by tests/test_oracle_golden.py against tests/golden/mas_kat.npz, which was produced by the
 * reference's own maximum_path().  Used as the CPU baseline of the MAS kernel and as the
 * checker for large shapes (the numpy twin in vits_oracle.py is too slow there).
 */
#include <stdint.h>

static float fmax2(float a, float b) { return a > b ? a : b; }

/* paths int32[b][ty][tx] (must be zero-filled), values float32[b][ty][tx] (modified in place,
 * like the reference's `values`), t_ys / t_xs int32[b]. */
void mas_oracle(int32_t* paths, float* values, const int32_t* t_ys, const int32_t* t_xs, int b,
                int ty, int tx) {
  const float max_neg_val = -1e9f;
  for (int i = 0; i < b; ++i) {
    int32_t* path = paths + (int64_t)i * ty * tx;
    float* value = values + (int64_t)i * ty * tx;
    const int t_y = t_ys[i], t_x = t_xs[i];
    float v_prev, v_cur;
    int index = t_x - 1;
    for (int y = 0; y < t_y; ++y) {
      int lo = t_x + y - t_y;
      if (lo < 0) lo = 0;
      int hi = t_x < y + 1 ? t_x : y + 1;
      for (int x = lo; x < hi; ++x) {
        if (x == y) v_cur = max_neg_val;
        else v_cur = value[(int64_t)(y - 1) * tx + x];
        if (x == 0) v_prev = (y == 0) ? 0.0f : max_neg_val;
        else v_prev = value[(int64_t)(y - 1) * tx + x - 1];
        value[(int64_t)y * tx + x] += fmax2(v_prev, v_cur);
      }
    }
    for (int y = t_y - 1; y >= 0; --y) {
      if (index < 0) break; /* t_x == 0: the reference would index column -1 */
      path[(int64_t)y * tx + index] = 1;
      /* value[y-1] with y == 0 wraps to the last row under numpy / numba indexing */
      const int yr = (y - 1 >= 0) ? (y - 1) : (ty - 1);
      if (index != 0 && (index == y || value[(int64_t)yr * tx + index] <
                                           value[(int64_t)yr * tx + index - 1]))
        index = index - 1;
    }
  }
}
