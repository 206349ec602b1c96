/*
 * rs_wave.h -- wave64 (CDNA4) building blocks of the cell kernel: DPP reductions and prefix scans, ballot-based
 * "lanes holding my value", exact small integer division, the glibc rand() ring held one word per lane.
 * Device-only fragment of rs_kernels.hip (same translation unit; also embedded for the hiprtc build).
 */
#ifndef RS_WAVE_H_
#define RS_WAVE_H_

#include "rs_device.h"

namespace {

/* a load from LDS that the compiler may neither cache in a register nor turn into a flat access: the flag words other waves
 * of the workgroup write (workgroup-scope relaxed atomic load = one ds_read_b32) */
__device__ __forceinline__ int rs_lds_load(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}


typedef RsMisc Misc;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

/* wave64 reductions on the DPP network (row_shr 1/2/4/8 inside rows of 16, then row_bcast15 and
 * row_bcast31 across rows); the total lands in lane 63 and is broadcast with v_readlane.
 * A lane masked off by the bank/row mask contributes the identity. */
#define RS_DPP_STEP(OP, ctrl, rmask, bmask) \
  v = OP(v, __builtin_amdgcn_update_dpp(identity, v, ctrl, rmask, bmask, false))
__device__ __forceinline__ int op_add(int a, int b) { return a + b; }
__device__ __forceinline__ int op_max(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int op_min(int a, int b) { return a < b ? a : b; }
#define RS_DEFINE_WAVE_REDUCE(NAME, OP, IDENT)                 \
  __device__ __forceinline__ int NAME(int v) {                 \
    const int identity = IDENT;                                \
    RS_DPP_STEP(OP, 0x111, 0xf, 0xf); /* row_shr:1 */          \
    RS_DPP_STEP(OP, 0x112, 0xf, 0xf); /* row_shr:2 */          \
    RS_DPP_STEP(OP, 0x114, 0xf, 0xe); /* row_shr:4 */          \
    RS_DPP_STEP(OP, 0x118, 0xf, 0xc); /* row_shr:8 */          \
    RS_DPP_STEP(OP, 0x142, 0xa, 0xf); /* row_bcast:15 */       \
    RS_DPP_STEP(OP, 0x143, 0xc, 0xf); /* row_bcast:31 */       \
    return __builtin_amdgcn_readlane(v, 63);                   \
  }
RS_DEFINE_WAVE_REDUCE(wave_sum, op_add, 0)
RS_DEFINE_WAVE_REDUCE(wave_max, op_max, (int)0x80000000)
RS_DEFINE_WAVE_REDUCE(wave_min, op_min, 0x7fffffff)

/* maximum over each 32-lane half of the wave, both halves at once (the ladder stops before row_bcast:31: lanes 31 and 63 hold
 * their half's maximum).  Every lane of the wave must take part. */
__device__ __forceinline__ int half_max(int v) {
  const int identity = (int)0x80000000;
  RS_DPP_STEP(op_max, 0x111, 0xf, 0xf); /* row_shr:1 */
  RS_DPP_STEP(op_max, 0x112, 0xf, 0xf); /* row_shr:2 */
  RS_DPP_STEP(op_max, 0x114, 0xf, 0xe); /* row_shr:4 */
  RS_DPP_STEP(op_max, 0x118, 0xf, 0xc); /* row_shr:8 */
  RS_DPP_STEP(op_max, 0x142, 0xa, 0xf); /* row_bcast:15 */
  const int lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
  return (threadIdx.x & 32) ? hi : lo;
}

/* lanes that hold the same BITS-bit value as this lane (valid lanes only): one ballot per bit */
template <int BITS>
struct BitBallots {
  unsigned long long valid, b[BITS];
  __device__ __forceinline__ void gather(int v, bool is_valid) {
    valid = __ballot(is_valid);
#pragma unroll
    for (int i = 0; i < BITS; ++i) b[i] = __ballot(((v >> i) & 1) != 0);
  }
  __device__ __forceinline__ unsigned long long lanes_with(int v) const {
    unsigned long long mk = valid;
#pragma unroll
    for (int i = 0; i < BITS; ++i) mk &= ((v >> i) & 1) ? b[i] : ~b[i];
    return mk;
  }
};

/* C integer division (truncation toward zero) for |a| < 2^20, 1 <= b <= 512: one correctly rounded
 * FP32 division instead of the ~35-instruction integer sequence; the fix-up makes it exact whatever
 * the rounding did. */
__device__ __forceinline__ int idiv_small(int a, int b) {
  int q = (int)((float)a / (float)b);
  int r = a - q * b;
  if (a >= 0) {
    if (r < 0) q--; else if (r >= b) q++;
  } else {
    if (r > 0) q++; else if (r <= -b) q--;
  }
  return q;
}

/* glibc TYPE_3 rand(): ring of 31 words held one per lane of one wave (lane l = r[l]); f, b uniform.
 * (glibc 2.35 stdlib/random_r.c __random_r; the reference draws from libc rand():
 *  downlink-transport-scheduler.cpp:490,511) */
struct WaveRng {
  uint32_t r; /* this lane's ring word */
  int f, b;   /* wave-uniform */
  __device__ __forceinline__ int next() {
    uint32_t vf = __builtin_amdgcn_readlane(r, f);
    uint32_t vb = __builtin_amdgcn_readlane(r, b);
    uint32_t v = vf + vb;
    r = ((int)(threadIdx.x & 63) == f) ? v : r;
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
    return (int)(v >> 1);
  }
  /* The next `count` (1..31) ring words at once: lane j < count returns x_j (rand() = x_j >> 1).  With the ring read in
   * age order old[0..30], x_j = old[j] + x_{j-3}, and x_{j-3} is an old word for j < 3: the new words are prefix sums of
   * the old ones along the three chains j mod 3, plus the chain's last old word old[28 + j mod 3].  Every lane of the
   * wave must call it (lane reads are ds_bpermute). */
  __device__ __forceinline__ uint32_t next_block(int count) {
    const int lane = (int)(threadIdx.x & 63);
    uint32_t old = r;
    if (f != 0) { /* wave-uniform: only after single draws moved the ring's start */
      int src = f + lane;
      src = src >= 31 ? src - 31 : src;
      old = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)r);
    }
    uint32_t s = lane < 31 ? old : 0u;
#pragma unroll
    for (int d = 3; d <= 24; d <<= 1) {
      const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - d) << 2, (int)s);
      if (lane >= d) s += t;
    }
    const int m3 = lane - 3 * ((lane * 43) >> 7); /* lane mod 3 for lane < 64 */
    const uint32_t nw = s + (uint32_t)__builtin_amdgcn_ds_bpermute((28 + m3) << 2, (int)old);
    /* the ring in age order again: the 31 - count youngest old words, then the new ones */
    if (count == 31) { /* wave-uniform */
      r = nw;
    } else {
      const uint32_t keep = (uint32_t)__builtin_amdgcn_ds_bpermute((lane + count) << 2, (int)old);
      const uint32_t fresh = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - (31 - count)) << 2, (int)nw);
      r = lane < 31 - count ? keep : fresh;
    }
    f = 0;
    b = 28;
    return nw;
  }
  /* Chain-major layout for streaming many blocks: age index j sits in lane 16*(j mod 3) + j/3, so each of the three chains
   * j mod 3 (11, 10, 10 words) lies in its own 16-lane row and the chain prefix sums are row-local DPP scans -- a block of
   * 31 words costs 4 DPP adds + 3 v_readlane, no LDS round trip.  cm_index() = the age index of this lane (-1: unused). */
  __device__ __forceinline__ static int cm_index() {
    const int lane = (int)(threadIdx.x & 63), row = lane >> 4, idx = lane & 15;
    return (row < 3 && idx < (row == 0 ? 11 : 10)) ? 3 * idx + row : -1;
  }
  __device__ __forceinline__ void to_chain_major() { /* from the ring (any f) */
    const int j = cm_index();
    int src = f + (j < 0 ? 0 : j);
    src = src >= 31 ? src - 31 : src;
    r = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)r);
    f = 0;
    b = 28;
  }
  __device__ __forceinline__ void to_age_order() { /* back to the ring with f = 0 */
    const int lane = (int)(threadIdx.x & 63);
    const int j = lane < 31 ? lane : 0;
    const int third = (j * 43) >> 7; /* j / 3 */
    r = (uint32_t)__builtin_amdgcn_ds_bpermute((16 * (j - 3 * third) + third) << 2, (int)r);
  }
  __device__ __forceinline__ uint32_t next_block_chain_major() { /* 31 new words, in the same layout */
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t old = cm_index() >= 0 ? r : 0u;
    int v = (int)old;
    const int identity = 0;
    RS_DPP_STEP(op_add, 0x111, 0xf, 0xf); /* row_shr:1 */
    RS_DPP_STEP(op_add, 0x112, 0xf, 0xf); /* row_shr:2 */
    RS_DPP_STEP(op_add, 0x114, 0xf, 0xe); /* row_shr:4 */
    RS_DPP_STEP(op_add, 0x118, 0xf, 0xc); /* row_shr:8 */
    /* chain c continues from old[28 + c]: age 28 = chain 1 index 9, 29 = chain 2 index 9, 30 = chain 0 index 10 */
    const uint32_t b0 = __builtin_amdgcn_readlane(old, 16 + 9), b1 = __builtin_amdgcn_readlane(old, 32 + 9),
                   b2 = __builtin_amdgcn_readlane(old, 10);
    const int row = lane >> 4;
    r = (uint32_t)v + (row == 0 ? b0 : row == 1 ? b1 : b2);
    return r;
  }
};

struct LdsArr {
  uint32_t* p;
  __device__ __forceinline__ uint32_t& operator[](int i) { return p[i]; }
};
struct LdsInt {
  int32_t* p;
  __device__ __forceinline__ int32_t& operator[](int i) { return p[i]; }
};


/* inclusive wave64 prefix sum (same DPP ladder as the reductions; every lane keeps its partial) */
__device__ __forceinline__ int wave_scan_incl(int v) {
  const int identity = 0;
  RS_DPP_STEP(op_add, 0x111, 0xf, 0xf); /* row_shr:1 */
  RS_DPP_STEP(op_add, 0x112, 0xf, 0xf); /* row_shr:2 */
  RS_DPP_STEP(op_add, 0x114, 0xf, 0xe); /* row_shr:4 */
  RS_DPP_STEP(op_add, 0x118, 0xf, 0xc); /* row_shr:8 */
  RS_DPP_STEP(op_add, 0x142, 0xa, 0xf); /* row_bcast:15 */
  RS_DPP_STEP(op_add, 0x143, 0xc, 0xf); /* row_bcast:31 */
  return v;
}

/* EXCLUSIVE prefix sum of values that live in lanes 0..7 only: shift by one lane, then three steps inside the first row of 16
 * lanes (the input is dead after the shift, so every step is one v_add_u32_dpp in place) */
__device__ __forceinline__ int wave_scan_excl8(int x) {
  const int identity = 0;
  int v = __builtin_amdgcn_update_dpp(identity, x, 0x111, 0xf, 0xf, true); /* row_shr:1, lane 0 <- 0 */
  RS_DPP_STEP(op_add, 0x111, 0xf, 0xf); /* row_shr:1 */
  RS_DPP_STEP(op_add, 0x112, 0xf, 0xf); /* row_shr:2 */
  RS_DPP_STEP(op_add, 0x114, 0xf, 0xe); /* row_shr:4 */
  return v;
}

/* inclusive wave64 prefix maximum */
__device__ __forceinline__ int wave_scan_max_incl(int v) {
  const int identity = (int)0x80000000;
  RS_DPP_STEP(op_max, 0x111, 0xf, 0xf);
  RS_DPP_STEP(op_max, 0x112, 0xf, 0xf);
  RS_DPP_STEP(op_max, 0x114, 0xf, 0xe);
  RS_DPP_STEP(op_max, 0x118, 0xf, 0xc);
  RS_DPP_STEP(op_max, 0x142, 0xa, 0xf);
  RS_DPP_STEP(op_max, 0x143, 0xc, 0xf);
  return v;
}

}  // namespace

#endif /* RS_WAVE_H_ */
