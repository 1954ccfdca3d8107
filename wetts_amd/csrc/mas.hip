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
                                                   float* __restrict__ values, int only_irregular) {
  extern __shared__ float rows[];  // [2][Tx] previous / current DP row
  const int b = blockIdx.x;
  const int t_y = t_ys[b], t_x = t_xs[b];
  const float* nc = neg_cent + (int64_t)b * Ty * Tx;
  float* val = values + (int64_t)b * Ty * Tx;
  int32_t* pth = path + (int64_t)b * Ty * Tx;
  const float max_neg_val = -1e9f;
  if (t_y <= 0 || t_x <= 0 || t_y > Ty || t_x > Tx) return;
  if (only_irregular && t_x <= t_y) return;  // answered by mas_wave_kernel

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


// ---------------------------------------------------------------------------------------------
// The fast form (round 3): ONE wave walks one utterance; lane l owns the C consecutive columns
// l*C .. l*C+C-1 (C = 1, 2, 4, 8, 16 => Tx <= 1024).
//   forward   the previous DP row lives in registers; the only cross-lane operand of a row step is
//             the left neighbour's last column, one DPP `wave_shr:1` move -- no LDS, no barrier in
//             the row loop.  neg_cent rows are fetched D rows ahead through a register ring.  The
//             comparison the backtrack will make at cell (y, x) -- value[y-1][x] < value[y-1][x-1],
//             or the forced step x == y (monotonic_align.py:52-57) -- has both operands in hand
//             during the forward step of that cell, so it is recorded there as ONE BIT per cell in
//             LDS (Ty*Tx/8 bytes: 12.8 KB at 800 x 128) and the DP table itself is never stored.
//             (Only in-band cells matter: the path stays inside the band, and an in-band cell's
//             two comparison operands are in-band cells of the previous row -- see DESIGN 3.7.)
//   backtrack a scalar walk over the bit table: one LDS read per 32/C rows (every lane reads its own
//             word of the row group, the next group's read is in flight during the walk), then
//             v_readlane + bit test per row; the column of every row goes to a u16 array in LDS.
//   output    all four waves of the block write the path rows (zeros included) as full coalesced rows, so
//             there is no memset pass and no scattered store.
// Bit-exact with the reference for t_x <= t_y (the only case MAS is defined for); utterances with
// t_x > t_y (the reference then reads wrapped / unwritten rows) are left to mas_kernel above.
// one v_max_f32: fmaxf() costs two extra canonicalising v_max per operand in the row recurrence (the values are
// never signalling NaNs; for ordinary NaNs v_max_f32 returns the other operand exactly like fmaxf)
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int C>
__global__ __launch_bounds__(256) void mas_wave_kernel(const float* __restrict__ neg_cent,
                                                       const int32_t* __restrict__ t_ys,
                                                       const int32_t* __restrict__ t_xs, int Ty, int Tx,
                                                       int32_t* __restrict__ path) {
  extern __shared__ uint32_t mas_lds[];
  constexpr int R = 32 / C;   // rows per bit word
  constexpr int RD = 4 * R;   // rows per block of the forward loop = depth of the row ring (128 loads in flight per lane)
  const int G = (Ty + RD - 1) / RD * 4;  // bit words per lane, rounded up to whole blocks
  uint32_t* bits = mas_lds;                                         // [G][64]
  unsigned short* idx = (unsigned short*)(mas_lds + (size_t)G * 64);  // [Ty] path column per row
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t_y = __builtin_amdgcn_readfirstlane(t_ys[b]);
  const int t_x = __builtin_amdgcn_readfirstlane(t_xs[b]);
  const bool valid = t_y > 0 && t_x > 0 && t_y <= Ty && t_x <= Tx && t_x <= t_y;
  const float* nc = neg_cent + (int64_t)b * Ty * Tx;
  int32_t* pth = path + (int64_t)b * Ty * Tx;
  const float max_neg_val = -1e9f;
  for (int y = threadIdx.x; y < Ty; y += blockDim.x) idx[y] = 0xFFFFu;
  __syncthreads();

  if (wave == 0 && valid) {
    const int x0 = lane * C;
    int offc[C];  // clamped column offsets (columns >= Tx read column Tx-1: loaded, never used)
#pragma unroll
    for (int c = 0; c < C; ++c) offc[c] = min(x0 + c, Tx - 1);
    // the row ring: RD rows x C columns = 128 loads in flight per lane -- one wave walks the utterance, so the ring
    // is all the memory-level parallelism there is: at 32 loads (the first straight-line version) a row still
    // took 230 ns, the round trip of a cold HBM line divided by 16 rows.  Every load below is issued
    // UNCONDITIONALLY (row index clamped) and every block of R rows is straight-line code: a load under a
    // branch makes the compiler wait with vmcnt(0) at the first use -- a memory round trip per row (what the
    // first version of this kernel did: 680 cycles per row).  Row base pointers are wave-uniform (scalar
    // registers), the per-lane part of an address is one 32-bit offset.
    float ring[RD][C];
    const int last = t_y - 1;
#pragma unroll
    for (int d = 0; d < RD; ++d) {
      const float* rowp = nc + (int64_t)min(d, last) * Tx;
#pragma unroll
      for (int c = 0; c < C; ++c) ring[d][c] = rowp[offc[c]];
    }
    float prev[C];
#pragma unroll
    for (int c = 0; c < C; ++c) prev[c] = 0.f;
    // column 0 never steps left (the backtrack tests index != 0 first): its bit is masked out once per row
    const uint32_t lanemask = (lane == 0) ? ~1u : ~0u;
    for (int yb = 0; yb < t_y; yb += RD) {  // rows >= t_y of the last block have an empty band: they change nothing
      uint32_t word = 0;
#pragma unroll
      for (int d = 0; d < RD; ++d) {
        const int y = yb + d;
        float raw[C];
#pragma unroll
        for (int c = 0; c < C; ++c) raw[c] = ring[d][c];
        {  // refill this ring slot with row y + RD
          const float* rowp = nc + (int64_t)min(y + RD, last) * Tx;
#pragma unroll
          for (int c = 0; c < C; ++c) ring[d][c] = rowp[offc[c]];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the refill where it is: RD rows ahead of its use
        // value[y-1][x-1] of the lane's first column: the left neighbour's last column of the previous row, one
        // DPP move; lane 0 has no neighbour and keeps `old` = the reference's boundary value for x == 0
        const float edge = (y == 0) ? 0.f : max_neg_val;
        const float left = __int_as_float(__builtin_amdgcn_update_dpp(
            __float_as_int(edge), __float_as_int(prev[C - 1]), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
        float cur[C];
        uint32_t rowbits = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const int x = x0 + c;
          const float pl = (c == 0) ? left : prev[c - 1];  // value[y-1][x-1]  (v_prev)
          const float pc = prev[c];                        // value[y-1][x]
          const bool diag = (x == y);
          const float v_cur = diag ? max_neg_val : pc;
          // every cell is updated, also outside the band max(0, t_x+y-t_y) <= x < min(t_x, y+1) (where the
          // reference leaves the raw score): an in-band cell's two operands are in-band cells of the previous row
          // (DESIGN 3.7), so what the others hold never reaches the path -- and the select costs 3 VALU per cell
          cur[c] = raw[c] + vmax(pl, v_cur);
          rowbits |= (diag || (pc < pl)) ? (1u << c) : 0u;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) prev[c] = cur[c];
        word |= (rowbits & lanemask) << ((d % R) * C);
        if (d % R == R - 1) {
          bits[(size_t)((yb + d) / R) * 64 + lane] = word;
          word = 0;
        }
      }
    }
    // the wave's own LDS writes are ordered before its reads by the compiler's waitcnt; no other wave reads
    int index = t_x - 1;
    int g = (t_y - 1) / R;
    uint32_t w = bits[(size_t)g * 64 + lane];
    for (; g >= 0; --g) {
      const uint32_t wn = bits[(size_t)max(g - 1, 0) * 64 + lane];  // next group's word, in flight
      const int y_top = min(t_y - 1, g * R + R - 1);
      for (int y = y_top; y >= g * R; --y) {
        if (lane == 0) idx[y] = (unsigned short)index;
        const uint32_t ws = (uint32_t)__builtin_amdgcn_readlane((int)w, index / C);
        const int bit = (ws >> ((y % R) * C + (index % C))) & 1u;
        index -= (index != 0) ? bit : 0;
      }
      w = wn;
    }
  }
  __syncthreads();

  // path rows, zeros included: every wave writes whole rows (coalesced)
  for (int y = wave; y < Ty; y += (int)(blockDim.x >> 6)) {
    const int col = idx[y];  // 0xFFFF: no path cell in this row
    for (int xb = 0; xb < Tx; xb += 64 * C) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int x = xb + c * 64 + lane;  // stride-64 columns: each store instruction is one contiguous run
        if (x < Tx) pth[(int64_t)y * Tx + x] = (x == col) ? 1 : 0;
      }
    }
  }
}

template <int C>
static void launch_mas_wave(const float* neg_cent, const int32_t* t_ys, const int32_t* t_xs, int B, int Ty,
                            int Tx, int32_t* path, size_t lds, hipStream_t s) {
  hipLaunchKernelGGL((mas_wave_kernel<C>), dim3(B), dim3(256), lds, s, neg_cent, t_ys, t_xs, Ty, Tx, path);
}

// utterances with t_x > t_y: the reference's arithmetic there (wrapped row reads) is reproduced by the
// general kernel; `only_irregular` makes it skip the utterances the wave kernel has already answered
int32_t k_mas(const float* neg_cent, const int32_t* t_ys, const int32_t* t_xs, int B, int Ty,
              int Tx, int32_t* path, float* values, hipStream_t s) {
  if (B == 0 || Ty == 0 || Tx == 0) return WETTS_OK;
  int threads = Tx >= 1024 ? 1024 : ((Tx + 63) / 64) * 64;
  size_t lds = (size_t)2 * Tx * sizeof(float);
  WETTS_REQUIRE(lds <= 64 * 1024, "MAS: Tx=%d too large for the LDS row buffers", Tx);
  // wave kernel: C columns per lane, bit table + row index array in LDS
  int C = 1;
  while (C * 64 < Tx) C *= 2;
  const int R = C <= 32 ? 32 / C : 0;
  const size_t wave_lds = R ? ((size_t)((Ty + 4 * R - 1) / (4 * R)) * 4 * 64 * 4 + (size_t)Ty * 2 + 16) : 0;
  bool fast = C <= 16 && wave_lds <= 150 * 1024 && Ty < 0xFFFF && Tx < 0xFFFF;
  if (fast && wave_lds > 64 * 1024) {  // > 64 KB of dynamic LDS needs the opt-in, per device; refused => plain kernel
    static signed char opt_in[5][64] = {};
    switch (C) {
      case 1: fast = lds_opt_in((const void*)mas_wave_kernel<1>, opt_in[0]); break;
      case 2: fast = lds_opt_in((const void*)mas_wave_kernel<2>, opt_in[1]); break;
      case 4: fast = lds_opt_in((const void*)mas_wave_kernel<4>, opt_in[2]); break;
      case 8: fast = lds_opt_in((const void*)mas_wave_kernel<8>, opt_in[3]); break;
      default: fast = lds_opt_in((const void*)mas_wave_kernel<16>, opt_in[4]); break;
    }
  }
  if (fast) {
    switch (C) {
      case 1: launch_mas_wave<1>(neg_cent, t_ys, t_xs, B, Ty, Tx, path, wave_lds, s); break;
      case 2: launch_mas_wave<2>(neg_cent, t_ys, t_xs, B, Ty, Tx, path, wave_lds, s); break;
      case 4: launch_mas_wave<4>(neg_cent, t_ys, t_xs, B, Ty, Tx, path, wave_lds, s); break;
      case 8: launch_mas_wave<8>(neg_cent, t_ys, t_xs, B, Ty, Tx, path, wave_lds, s); break;
      default: launch_mas_wave<16>(neg_cent, t_ys, t_xs, B, Ty, Tx, path, wave_lds, s); break;
    }
    WETTS_LAUNCH_CHECK();
  } else {
    WETTS_HIP_CHECK(hipMemsetAsync(path, 0, (size_t)B * Ty * Tx * sizeof(int32_t), s));
  }
  hipLaunchKernelGGL(mas_kernel, dim3(B), dim3(threads), lds, s, neg_cent, t_ys, t_xs, Ty, Tx, path,
                     values, fast ? 1 : 0);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
