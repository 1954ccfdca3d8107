// Fused ResBlock1 pair at float32:   out = (x + c2(lrelu(c1(lrelu(x)))) [+ out]) / div
// (reference decoders.py:157-170, one (c1, c2) iteration of ResBlock1.forward) in ONE kernel on
// v_mfma_f32_32x32x2_f32 -- the f32 sibling of resblock16.hip.
//
// What it buys over two conv_mfma_kernel launches (profiles/r01_conv32_fused_pair.txt):
//   * the WHOLE channel depth of the x tile is staged once, before any MFMA: the matrix loop then
//     carries only the weight-fragment loads, so the in-order vmcnt counter no longer couples an
//     HBM-latency staging load to every A-fragment wait (the ~14 % the chunked kernel loses,
//     profiles/r01_conv_ablation_compiletime.txt), and the A prefetch waits are exact;
//   * the intermediate ft never leaves the CU (LDS, written over the x tile) and the residual is
//     an L2 hit: 2 tensor passes through HBM instead of 5 -- the C=32/64, k=3 pairs sit below the
//     f32 ridge (16-32 flop/B) and were HBM-bound.
//
// Geometry (activations stay channel-first f32 [B][C][T]):
//   LDS tile [C][Wp] floats, columns = times tx0 .. tx0+Wp, tx0 = n0 - h2 - h1,
//   h1 = (k-1)/2*d (c1 halo), h2 = (k-1)/2 (c2 halo), Wp = roundup4(NTC + 2*h1).
//   c1 computes NTC columns t = n0-h2+c; its output (+b1, lrelu, zero outside [0,T)) is written
//   to columns c of the same rows; c2 computes columns t' = n0+c' and keeps c' < NTO = NTC-2*h2.
// Operation order and rounding points equal the two-launch path (same weight packing, same
// group order), so results are bit-identical to it.
#include "common.h"
#include "resblock32.h"

namespace wetts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };  // dword-aligned 16-byte load

// RB2 = true: the ResBlock2 chain (decoders.py:205-214)  t1 = x + c1(lrelu(x)),
// out = t1 + c2(lrelu(t1)) with c2 at dilation p.dil2.  c2 uses the SAME column -> time mapping as
// c1 (column c = time n0-h2+c), so the raw t1 a lane needs as c2's residual is the value already in
// its own accumulators: c2 simply keeps accumulating.  lrelu(t1) goes to the tile shifted by h2
// columns (c2 reads column c + tap*dil2); valid outputs are the middle columns [h2, NTC-h2).
// (A first version kept the tile raw and applied leaky-relu at the B read: the two VALU ops in
// front of every MFMA made it slower than two launches.)
template <int C, bool RB2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void resblock_pair32_kernel(const ResPair32Params p) {
  constexpr int WM = C / 32, WN = 4 / WM, NB = 4;
  constexpr int NTC = 32 * NB * WN;
  constexpr int NCH = C / kConvCK;  // 16-channel chunks of the packed weights
  constexpr int RPW = C / 4;        // staged rows per wave
  constexpr int QN = (NTC + RESPAIR32_MAX_SPAN + 255) / 256;  // 16-byte pieces per lane per row
  constexpr int RB = RPW;  // rows per staging batch: the whole tile in flight at once (one latency)

  extern __shared__ __attribute__((aligned(16))) float smem_p[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;

  const int dil2 = RB2 ? p.dil2 : 1;
  const int h2 = (p.ktaps - 1) / 2 * dil2, h1 = (p.ktaps - 1) / 2 * p.dil;
  const int NTO = NTC - 2 * h2;
  const int Wp = p.Wp;
  int bid = blockIdx.x;
  {  // XCD-aware tile order (see resblock16.hip)
    const int per = (p.nblocks + 7) >> 3;
    bid = (bid & 7) * per + (bid >> 3);
    if (bid >= p.nblocks) return;
  }
  const int ntile = bid % p.ntiles;
  const int b = bid / p.ntiles;
  const int n0 = ntile * NTO;
  const int tx0 = n0 - h2 - h1;

  const float* xb = p.x + (int64_t)b * C * p.T;
  float* ob = p.out + (int64_t)b * C * p.T;

  // ---- A streams: [m-block][group][lane][4 k-steps], group = (chunk*ktaps + tap)*2 + hp ------
  const int G = NCH * p.ktaps * 2;
  const float4* abase1 = reinterpret_cast<const float4*>(p.wpk1) + ((int64_t)wm * G) * 64 + lane;
  const float4* abase2 = reinterpret_cast<const float4*>(p.wpk2) + ((int64_t)wm * G) * 64 + lane;
  float4 aa[2];  // A fragments of the groups g (aa[g & 1]) and g + 1, prefetched two groups ahead
  aa[0] = abase1[0];
  aa[1] = abase1[64];

  // Raw residual, requested first so that it is the oldest load in flight.  ResBlock1: x at c2's
  // output columns (time n0 + col), consumed after c1.  ResBlock2: x at c1's columns (time
  // n0 - h2 + col), which initialises c1's accumulators.
  const int co_blk = wm * 32;
  const int wcol = wn * (32 * NB) + (lane & 31);
  float rres[NB][16];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int col = wcol + 32 * j;
    const int t = RB2 ? n0 - h2 + col : n0 + col;
    const bool ok = (RB2 ? t >= 0 : col < NTO) && t < p.T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = 0.f;
      if (ok) v = xb[(int64_t)(co_blk + (r & 3) + 8 * (r >> 2) + 4 * half) * p.T + t];
      rres[j][r] = v;
    }
  }

  // ---- 1. stage lrelu(x): wave w takes rows w, w+4, ...; a lane takes 16-byte pieces --------
  {
    const int ppr = Wp >> 2;  // pieces per row
#pragma unroll 1
    for (int r0 = 0; r0 < RPW; r0 += RB) {
      float4 st[RB][QN];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int row = wave + 4 * (r0 + r);
        const float* xr = xb + (int64_t)row * p.T;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int seg = lane + 64 * q;
          const int t = tx0 + 4 * seg;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (seg < ppr) {
            if (t >= 0 && t + 3 < p.T) {
              const f4u u = *reinterpret_cast<const f4u*>(xr + t);
              v = make_float4(u.x, u.y, u.z, u.w);
            } else {  // sequence edge: element-wise, zero outside [0, T)
              if (t >= 0 && t < p.T) v.x = xr[t];
              if (t + 1 >= 0 && t + 1 < p.T) v.y = xr[t + 1];
              if (t + 2 >= 0 && t + 2 < p.T) v.z = xr[t + 2];
              if (t + 3 >= 0 && t + 3 < p.T) v.w = xr[t + 3];
            }
          }
          st[r][q] = v;
        }
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int row = wave + 4 * (r0 + r);
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int seg = lane + 64 * q;
          if (seg < ppr) {
            float4 v = st[r][q];
            v.x = v.x > 0.f ? v.x : v.x * p.slope;
            v.y = v.y > 0.f ? v.y : v.y * p.slope;
            v.z = v.z > 0.f ? v.z : v.z * p.slope;
            v.w = v.w > 0.f ? v.w : v.w * p.slope;
            *reinterpret_cast<float4*>(smem_p + (size_t)row * Wp + 4 * seg) = v;
          }
        }
      }
    }
  }
  __syncthreads();

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const float* bcol = smem_p + (size_t)half * Wp + wcol;
  if (RB2) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = rres[j][r];
  }

  // One group = 16 MFMAs: k-steps s = 0..3 are channels chunk*16 + hp*8 + 2s + half, each feeding
  // the NB column blocks.  The B fragments are software-pipelined ONE k-step ahead in registers
  // (bv[parity][j]): without this the compiler places every ds_read directly in front of the MFMAs
  // that use it (`ds_read2_b32; s_waitcnt lgkmcnt(0); 2 x v_mfma`), i.e. one exposed LDS round trip
  // per 128 MFMA cycles -- at the two waves per SIMD this kernel's LDS tile allows, that alone held
  // the MFMA + LDS part of a C = 32, k = 3 pair at 89 TF/s (profiles/r02_pair32_ablation.txt).
  // sched_barriers pin the order: [A prefetch] [B reads of step s+1] [MFMAs of step s].
  // The A prefetch is issued unconditionally (group G exists: next m-block / zero tail of the
  // packed buffer) a whole group ahead, so the compiler's vmcnt waits are exact (a 4-slot ring with
  // 3 groups of cover measured 2-4 % slower).
  auto conv_loop = [&](const float4* abase, int dil) {
    const int G = NCH * p.ktaps * 2;
    float bv[2][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[0][j] = bcol[32 * j];  // group 0, step 0 (chunk 0, tap 0, hp 0)
    // one (chunk, tap) iteration = two groups (hp = 0, 1): aa[0] / aa[1] without dynamic indexing
    auto group = [&](float4& areg, const float4* anext, const float* cur, const float* nxt) {
      const float4 av = areg;
      areg = *anext;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* src = s < 3 ? cur + (size_t)(2 * (s + 1)) * Wp : nxt;
#pragma unroll
        for (int j = 0; j < NB; ++j) bv[(s + 1) & 1][j] = src[32 * j];
        __builtin_amdgcn_sched_barrier(0);
        const float a = s == 0 ? av.x : s == 1 ? av.y : s == 2 ? av.z : av.w;
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[s & 1][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    int g = 0;
    for (int chunk = 0; chunk < NCH; ++chunk) {
      for (int tap = 0; tap < p.ktaps; ++tap) {
        const float* r0 = bcol + (size_t)(chunk * kConvCK) * Wp + tap * dil;  // hp = 0 rows
        const float* r1 = r0 + (size_t)8 * Wp;                                // hp = 1 rows
        // first rows of the next (chunk, tap); the very last prefetch re-reads this group's rows
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == p.ktaps) { ntap = 0; ++nchunk; }
        const float* rn = (g + 2 < G) ? bcol + (size_t)(nchunk * kConvCK) * Wp + ntap * dil : r1;
        group(aa[0], abase + (int64_t)(g + 2) * 64, r0, r1);
        group(aa[1], abase + (int64_t)(g + 3) * 64, r1, rn);
        g += 2;
      }
    }
  };

  // ---- 2. c1 ---------------------------------------------------------------------------------
  conv_loop(abase1, p.dil);
  aa[0] = abase2[0];  // c2's first two groups; they land during step 3
  aa[1] = abase2[64];

  // ---- 3. ft = lrelu(c1 + b1) over the x tile (zero outside [0,T): c2 pads ITS input) ---------
  {
    float bia[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bia[r] = p.bias1[co_blk + (r & 3) + 8 * (r >> 2) + 4 * half];
    __syncthreads();  // every wave has finished reading lrelu(x)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = wcol + 32 * j;
      const int t = n0 - h2 + col;
      const bool inside = t >= 0 && t < p.T;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[j][r] + bia[r];
        if (RB2) acc[j][r] = v;  // raw t1: c2's residual, already in place
        v = v > 0.f ? v : v * p.slope;
        smem_p[(size_t)(co_blk + (r & 3) + 8 * (r >> 2) + 4 * half) * Wp + col + (RB2 ? h2 : 0)] =
            inside ? v : 0.f;
      }
    }
  }

  // ---- 4. c2: accumulator = raw residual (+ running MRF sum) -----------------------------------
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int col = wcol + 32 * j;
    const int t = RB2 ? n0 - h2 + col : n0 + col;
    // the running MRF sum: a column block's 16 rows requested in one batch from a column clamped into the utterance
    // (columns outside the stored range are never stored), not one load / wait / add round trip per element
    // (tools/isa_scan.py; the f32 conv kernel lost the same shape in round 5)
    float os[16];
    if (p.accum) {
      const int tc = t < 0 ? 0 : (t >= p.T ? p.T - 1 : t);
#pragma unroll
      for (int r = 0; r < 16; ++r) os[r] = ob[(int64_t)(co_blk + (r & 3) + 8 * (r >> 2) + 4 * half) * p.T + tc];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = RB2 ? acc[j][r] : rres[j][r];
      if (p.accum) v += os[r];
      acc[j][r] = v;
    }
  }
  __syncthreads();  // ft complete
  conv_loop(abase2, dil2);

  // ---- 5. epilogue -----------------------------------------------------------------------------
  float bia[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bia[r] = p.bias2[co_blk + (r & 3) + 8 * (r >> 2) + 4 * half];
  auto store_all = [&](auto fin) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = wcol + 32 * j;
      const int t = RB2 ? n0 - h2 + col : n0 + col;
      if (RB2 ? (col < h2 || col >= NTC - h2 || t < 0) : col >= NTO) continue;
      if (t >= p.T) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ob[(int64_t)(co_blk + (r & 3) + 8 * (r >> 2) + 4 * half) * p.T + t] = fin(acc[j][r] + bia[r]);
    }
  };
  const float dv = p.out_div, dinv = 1.f / p.out_div;  // the MRF mean (common.h: mrf_div)
  if (dv == 1.f) store_all([](float v) { return v; });
  else if (mrf_div_fast(dv)) store_all([=](float v) { return div_small_const(v, dv, dinv); });
  else store_all([=](float v) { return v / dv; });
}

template <int C, bool RB2>
static int32_t launch_pair32(const ResPair32Params& p0, hipStream_t stream) {
  constexpr int WM = C / 32, WN = 4 / WM, NTC = 128 * WN;
  ResPair32Params p = p0;
  const int h2 = (p.ktaps - 1) / 2 * (RB2 ? p.dil2 : 1), h1 = (p.ktaps - 1) / 2 * p.dil;
  const int NTO = NTC - 2 * h2;
  WETTS_REQUIRE(NTO > 0, "second conv's halo exceeds the tile");
  p.ntiles = cdiv(p.T, NTO);
  // the tile also backs c2's reads of its (discarded) last columns: NTC + 2*max(h1, h2) wide
  p.Wp = (NTC + 2 * (h1 > h2 ? h1 : h2) + 3) & ~3;
  const int64_t nb = (int64_t)p.ntiles * p.B;
  if (nb <= 0) return WETTS_OK;
  WETTS_REQUIRE(nb < (1ll << 30), "resblock grid too large");
  p.nblocks = (int)nb;
  const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);
  const size_t lds = (size_t)C * p.Wp * sizeof(float);
  int dev = 0;
  (void)hipGetDevice(&dev);
  static bool attr_done[64] = {};  // per device: tiles above the default 64 KB dynamic-LDS limit
  if (dev >= 0 && dev < 64 && !attr_done[dev]) {
    WETTS_HIP_CHECK(hipFuncSetAttribute((const void*)resblock_pair32_kernel<C, RB2>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((resblock_pair32_kernel<C, RB2>), dim3(grid), dim3(256), lds, stream, p);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

bool resblock_pair32_supported(const PackedConv& c1, const PackedConv& c2, int max_lds_bytes) {
  const int C = c1.Cin;
  if (!(C == 32 || C == 64 || C == 128)) return false;
  if (c1.Cout != C || c2.Cin != C || c2.Cout != C || c1.up || c2.up) return false;
  if (c1.M != C || c2.M != C) return false;
  if (c1.ktaps != c2.ktaps || (c1.ktaps & 1) == 0 || c2.dil != 1) return false;
  if (c1.pad != (c1.ktaps - 1) / 2 * c1.dil || c2.pad != (c2.ktaps - 1) / 2) return false;
  if ((c1.ktaps - 1) * c1.dil > RESPAIR32_MAX_SPAN) return false;
  const int NTC = 128 * (4 / (C / 32));
  const int Wp = (NTC + (c1.ktaps - 1) * c1.dil + 3) & ~3;
  return (int64_t)C * Wp * 4 <= max_lds_bytes;
}

int32_t launch_resblock_pair32(const PackedConv& c1, const PackedConv& c2, ResPair32Params p,
                               hipStream_t stream) {
  WETTS_REQUIRE(resblock_pair32_supported(c1, c2, 160 * 1024),
                "resblock pair shape not supported by the fused f32 kernel");
  WETTS_REQUIRE(c1.wpk && c2.wpk, "conv weight not packed");
  p.wpk1 = c1.wpk; p.bias1 = c1.bias;
  p.wpk2 = c2.wpk; p.bias2 = c2.bias;
  WETTS_REQUIRE(p.bias1 && p.bias2, "resblock convs carry a bias");
  p.ktaps = c1.ktaps;
  p.dil = c1.dil;
  p.dil2 = 1;
  switch (c1.Cin) {
    case 32: return launch_pair32<32, false>(p, stream);
    case 64: return launch_pair32<64, false>(p, stream);
    default: return launch_pair32<128, false>(p, stream);
  }
}

// ResBlock2 (decoders.py:205-214): both convs carry the residual; c2 keeps its own dilation
bool resblock2_chain32_supported(const PackedConv& c1, const PackedConv& c2, int max_lds_bytes,
                                 int max_waste_pct) {
  const int C = c1.Cin;
  if (!(C == 32 || C == 64 || C == 128)) return false;
  if (c1.Cout != C || c2.Cin != C || c2.Cout != C || c1.up || c2.up) return false;
  if (c1.M != C || c2.M != C) return false;
  if (c1.ktaps != c2.ktaps || (c1.ktaps & 1) == 0) return false;
  if (c1.pad != (c1.ktaps - 1) / 2 * c1.dil || c2.pad != (c2.ktaps - 1) / 2 * c2.dil) return false;
  const int span = (c1.ktaps - 1) * (c1.dil > c2.dil ? c1.dil : c2.dil);
  if (span > RESPAIR32_MAX_SPAN) return false;
  const int NTC = 128 * (4 / (C / 32));
  const int lost = (c2.ktaps - 1) * c2.dil;  // columns of the tile the second conv cannot produce
  if (lost * 100 > max_waste_pct * NTC) return false;
  const int Wp = (NTC + span + 3) & ~3;
  return (int64_t)C * Wp * 4 <= max_lds_bytes;
}

int32_t launch_resblock2_chain32(const PackedConv& c1, const PackedConv& c2, ResPair32Params p,
                                 hipStream_t stream) {
  WETTS_REQUIRE(resblock2_chain32_supported(c1, c2, 160 * 1024, 100),
                "ResBlock2 shape not supported by the fused f32 kernel");
  WETTS_REQUIRE(c1.wpk && c2.wpk && c1.bias && c2.bias, "conv weight not packed");
  p.wpk1 = c1.wpk; p.bias1 = c1.bias;
  p.wpk2 = c2.wpk; p.bias2 = c2.bias;
  p.ktaps = c1.ktaps;
  p.dil = c1.dil;
  p.dil2 = c2.dil;
  switch (c1.Cin) {
    case 32: return launch_pair32<32, true>(p, stream);
    case 64: return launch_pair32<64, true>(p, stream);
    default: return launch_pair32<128, true>(p, stream);
  }
}

}  // namespace wetts
