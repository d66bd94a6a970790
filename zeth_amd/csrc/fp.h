// fp.h — BabyBear (P = 15*2^27 + 1) Montgomery arithmetic and the x^4 + 11 extension, host + gfx950 device.
// Stands in for risc0-core 3.0.0 src/field/baby_bear.rs (un-vendored; /root/reference/Cargo.lock:5338):
// Elem = u32 Montgomery word (< P), ExtElem = 4 Elems.  Reached from
// /root/reference/crates/host/src/lib.rs:137 via risc0_zkp::hal::Hal.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZKH_HD __host__ __device__ __forceinline__
#else
#define ZKH_HD inline
#endif

namespace zkh {

constexpr uint32_t P = 2013265921u;        // 0x78000001
constexpr uint32_t PINV = 0x88000001u;     // P^-1 mod 2^32
constexpr uint32_t NEG_PINV = 0x77ffffffu; // -P^-1 mod 2^32
constexpr uint32_t R1 = 268435454u;        // 2^32 mod P  (Montgomery 1)
constexpr uint32_t R2 = 1172168163u;       // 2^64 mod P
constexpr uint32_t INVALID = 0xffffffffu;

struct Fp {
    uint32_t v;   // Montgomery form, < P
    ZKH_HD Fp() : v(0) {}
    ZKH_HD explicit constexpr Fp(uint32_t raw, int) : v(raw) {}
    static ZKH_HD Fp raw(uint32_t r) { return Fp(r, 0); }
    static ZKH_HD Fp one() { return Fp(R1, 0); }
    static ZKH_HD Fp zero() { return Fp(0, 0); }
};

// Conditional corrections are written with the overflow builtins so that hipcc emits v_sub(rev)_co_u32 + v_cndmask_b32.
// Measured on gfx950 (tools/ubench_valu.hip, profiles/ubench_valu_r01.txt): v_add/v_sub/v_and/v_xor/v_ashrrev issue at
// ~64 T lane-ops/s, while v_min_u32, v_add3, every 32-bit multiply, v_mad_u64_u32 and all fp64 ops issue at ~37 T/s; a
// sub_co + cndmask pair costs ~2.1 add-slots against ~2.7 for sub + min.
// v_subrev_co_u32 + v_cndmask_b32: measured cheaper than v_sub_u32 + v_min_u32 (hash_rows 13.3 vs 13.8 ms)
// ZKH_REDUCE_MIN (a per-translation-unit build option, used by the generated eval_check kernels): the correction as
// v_sub + v_min_u32 instead of v_subrev_co + v_cndmask.  Same results; no VCC dependency, hence none of the s_nop hazard
// fillers that make up ~20 % of the instruction stream of straight-line field arithmetic.
ZKH_HD uint32_t reduce_once(uint32_t s) {     // s in [0, 2P) -> [0, P)
#if defined(ZKH_REDUCE_MIN)
    const uint32_t t = s - P;                 // wraps above s exactly when s < P
    return t < s ? t : s;
#else
    uint32_t t;
    const bool borrow = __builtin_usub_overflow(s, P, &t);
    return borrow ? s : t;
#endif
}
ZKH_HD uint32_t add_mod(uint32_t a, uint32_t b) { return reduce_once(a + b); }
ZKH_HD uint32_t sub_mod(uint32_t a, uint32_t b) {
#if defined(ZKH_REDUCE_MIN)
    const uint32_t t = a - b, u = t + P;      // a >= b: u = t + P > t;  a < b: t is huge and u wraps to a - b + P < t
    return u < t ? u : t;
#else
    uint32_t t;
    const bool borrow = __builtin_usub_overflow(a, b, &t);
    return borrow ? t + P : t;
#endif
}
// Montgomery reduction of a 64-bit product T < P * 2^32:  (T + m*P) / 2^32 with m = -T * P^-1 mod 2^32 makes the low word
// cancel, so ONE 64-bit multiply-add (v_mad_u64_u32 m, P, T) yields the quotient in its high register: two multiplies
// and one multiply-add per product instead of three multiplies and a subtract.  T + m*P < 2P * 2^32 < 2^64.
ZKH_HD uint32_t mont_reduce(uint64_t t) {
    const uint32_t m = (uint32_t)t * NEG_PINV;
    const uint64_t s = t + (uint64_t)m * P;
    return reduce_once((uint32_t)(s >> 32));
}
// Signed Montgomery product (Seiler): for |a|, |b| < P the result is in (-P, P) and congruent to a*b*2^-32 with NO
// correction step, so chains of products (the x^7 s-box) only canonicalise once at the end.  |t + m*P| < 2^63.
// acc + a*b as an exact (wrapping) 64-bit sum of a signed 32x32 product.  On the device it is pinned to ONE
// v_mad_i64_i32: left to itself hipcc hoists the sign extensions of loop-invariant operands and then expands the
// product as a generic 64x64 multiply (v_mad_u64_u32 + 2 v_mul_lo_u32 + v_add3_u32).  `_k`: b is wave-uniform (SGPR).
ZKH_HD int64_t mad_i64(int32_t a, int32_t b, int64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d; uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(acc));
    return d;
#else
    return (int64_t)((uint64_t)acc + (uint64_t)((int64_t)a * b));
#endif
}
ZKH_HD int64_t mad_i64_k(int32_t a, int32_t b_uniform, int64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d; uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(b_uniform), "v"(acc));
    return d;
#else
    return (int64_t)((uint64_t)acc + (uint64_t)((int64_t)a * b_uniform));
#endif
}
ZKH_HD int64_t mul_i64(int32_t a, int32_t b) {           // exact signed 32x32 -> 64 product: v_mad_i64_i32 with a literal 0 addend
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d; uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
    return d;
#else
    return (int64_t)a * b;
#endif
}
ZKH_HD int32_t smont_reduce(int64_t t) {                // any exact signed sum with |t| < P*2^31  ->  t*2^-32 in (-P, P)
    const int32_t m = (int32_t)((uint32_t)t * NEG_PINV);
    return (int32_t)(mad_i64_k(m, (int32_t)P, t) >> 32); // t + m*P: low 32 bits are zero
}
ZKH_HD int32_t smont(int32_t a, int32_t b) {
    const int64_t t = (int64_t)a * b;
    const int32_t m = (int32_t)((uint32_t)t * NEG_PINV);
    const int64_t s = t + (int64_t)m * (int64_t)P;      // low 32 bits are zero
    return (int32_t)(s >> 32);
}
// [-P, P) -> [0, P): for x < 0 the unsigned x + P wraps below x, so the unsigned minimum picks the right one
// (v_add_u32 + v_min_u32; measured 1 % faster in hash_rows than v_ashrrev + v_and + v_add).
ZKH_HD uint32_t canon(int32_t x) { const uint32_t u = (uint32_t)x, v = u + P; return v < u ? v : u; }
ZKH_HD int32_t center(uint32_t x) { return (int32_t)(x - (x > (P - 1) / 2 ? P : 0u)); }  // [0, P) -> [-(P-1)/2, (P-1)/2]
ZKH_HD uint32_t mul_mod(uint32_t a, uint32_t b) { return mont_reduce((uint64_t)a * b); }

ZKH_HD Fp operator+(Fp a, Fp b) { return Fp::raw(add_mod(a.v, b.v)); }
ZKH_HD Fp operator-(Fp a, Fp b) { return Fp::raw(sub_mod(a.v, b.v)); }
ZKH_HD Fp operator*(Fp a, Fp b) { return Fp::raw(mul_mod(a.v, b.v)); }
ZKH_HD Fp operator-(Fp a) { return Fp::raw(a.v ? P - a.v : 0); }
ZKH_HD bool operator==(Fp a, Fp b) { return a.v == b.v; }
ZKH_HD Fp& operator+=(Fp& a, Fp b) { a = a + b; return a; }
ZKH_HD Fp& operator-=(Fp& a, Fp b) { a = a - b; return a; }
ZKH_HD Fp& operator*=(Fp& a, Fp b) { a = a * b; return a; }

ZKH_HD Fp fp_encode(uint32_t x) { return Fp::raw(mul_mod(R2, x % P)); }
ZKH_HD uint32_t fp_decode(Fp a) { return mul_mod(1u, a.v); }
ZKH_HD Fp fp_pow(Fp a, uint64_t e) {
    Fp r = Fp::one();
    while (e) { if (e & 1) r = r * a; a = a * a; e >>= 1; }
    return r;
}
ZKH_HD Fp fp_inv(Fp a) { return fp_pow(a, P - 2); }

constexpr uint32_t NBETA_CANON = P - 11;     // x^4 = -11
// Montgomery forms of 11 and -11 (computed once on host and checked in tests): 11 * 2^32 mod P
constexpr uint32_t BETA_M = (uint32_t)((11ull << 32) % P);
constexpr uint32_t NBETA_M = P - BETA_M;

struct Fp4 {
    Fp c[4];
    ZKH_HD Fp4() {}
    ZKH_HD Fp4(Fp a, Fp b, Fp d, Fp e) { c[0] = a; c[1] = b; c[2] = d; c[3] = e; }
    ZKH_HD explicit Fp4(Fp a) { c[0] = a; c[1] = c[2] = c[3] = Fp::zero(); }
    static ZKH_HD Fp4 zero() { return Fp4(Fp::zero()); }
    static ZKH_HD Fp4 one() { return Fp4(Fp::one()); }
};
ZKH_HD Fp4 operator+(Fp4 a, Fp4 b) { return Fp4(a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2], a.c[3] + b.c[3]); }
ZKH_HD Fp4 operator-(Fp4 a, Fp4 b) { return Fp4(a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2], a.c[3] - b.c[3]); }
ZKH_HD Fp4 operator*(Fp4 a, Fp b) { return Fp4(a.c[0] * b, a.c[1] * b, a.c[2] * b, a.c[3] * b); }
ZKH_HD bool operator==(Fp4 a, Fp4 b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3]; }

// Schoolbook product mod x^4 + 11 with lazy accumulation: 64-bit partial products (each < P^2) are summed in a
// v_mad_u64_u32 chain and reduced ONCE per output coefficient (7 reductions + 19 wide multiplies instead of
// 19 full Montgomery products).  Up to 4 products fit: 4*P^2 < 2^64 and 4*P^2 < 2*P*2^32, so one conditional
// subtraction of P from the high word brings the sum under the P*2^32 bound mont_reduce needs.
ZKH_HD uint32_t mont_reduce_wide(uint64_t t) {   // any t < 2*P*2^32
    const uint32_t hi = reduce_once((uint32_t)(t >> 32));
    return mont_reduce(((uint64_t)hi << 32) | (uint32_t)t);
}
// Lazy forms for straight-line generated code (circuits/codegen.py, Plan.find_lazy): a value whose every consumer is a
// product may stay in [0, 2P).  With one factor below 2P and the other below P the product is below 2 P^2 < P 2^32, and
// the Montgomery step's uncorrected output (T + m P) / 2^32 < 2P again fits the same form.
ZKH_HD uint32_t mont_reduce_lazy(uint64_t t) {        // t < P 2^32  ->  [0, 2P)
    const uint32_t m = (uint32_t)t * NEG_PINV;
    return (uint32_t)((t + (uint64_t)m * P) >> 32);
}
ZKH_HD uint32_t mont_reduce_wide_lazy(uint64_t t) {   // t < 2 P 2^32  ->  [0, 2P)
    const uint32_t hi = reduce_once((uint32_t)(t >> 32));
    return mont_reduce_lazy(((uint64_t)hi << 32) | (uint32_t)t);
}
ZKH_HD uint32_t mul_lazy(uint32_t a, uint32_t b) { return mont_reduce_lazy((uint64_t)a * b); }   // a b < P 2^32
// Room in a 64-bit sum of products without reducing it: hi 2^32 + lo = hi R + lo (mod P), below 2^60 + 2^32.
ZKH_HD uint64_t fold_acc(uint64_t s) { return (s >> 32) * R1 + (uint32_t)s; }

// SIGNED twins for the generated eval_check kernels' running constraint sums (circuits/codegen.py SIGNED): the mix powers are read
// CENTRED (|p| <= (P-1)/2, as int32) and a lazy operand x in [0, 2P) enters as x - P in [-P, P) (the same residue), so every leaf
// is |p r| <= (P-1)/2 * P ~ P^2 / 2 whether its operand was reduced or not — four leaves fit the signed 64-bit sum (2^63 = 2.27 P^2)
// between folds, where the unsigned sums (4.55 P^2) held only two lazy ones (2 P^2 each).
// fold: s = hi 2^32 + lo with hi SIGNED and lo unsigned = hi R + lo (mod P); |result| < 2^31 R + 2^32 < 2^59.1
ZKH_HD int64_t fold_acc_s(int64_t s) { return mad_i64_k((int32_t)(s >> 32), (int32_t)R1, (int64_t)(uint32_t)s); }
// the canonical word of a signed sum |t| < P 2^31: the signed Montgomery step (uncorrected, in (-P, P)) and one conditional + P
ZKH_HD uint32_t smont_canon(int64_t t) { return canon(smont_reduce(t)); }

ZKH_HD Fp4 operator*(Fp4 a, Fp4 b) {
    const uint64_t a0 = a.c[0].v, a1 = a.c[1].v, a2 = a.c[2].v, a3 = a.c[3].v;
    const uint64_t b0 = b.c[0].v, b1 = b.c[1].v, b2 = b.c[2].v, b3 = b.c[3].v;
    const uint64_t h0 = mont_reduce_wide(a1 * b3 + a2 * b2 + a3 * b1);   // x^4 coefficient
    const uint64_t h1 = mont_reduce_wide(a2 * b3 + a3 * b2);             // x^5
    const uint64_t h2 = mont_reduce(a3 * b3);                            // x^6
    Fp4 r;
    r.c[0] = Fp::raw(mont_reduce_wide(a0 * b0 + NBETA_M * h0));
    r.c[1] = Fp::raw(mont_reduce_wide(a0 * b1 + a1 * b0 + NBETA_M * h1));
    r.c[2] = Fp::raw(mont_reduce_wide(a0 * b2 + a1 * b1 + a2 * b0 + NBETA_M * h2));
    r.c[3] = Fp::raw(mont_reduce_wide(a0 * b3 + a1 * b2 + a2 * b1 + a3 * b0));
    return r;
}
ZKH_HD Fp4& operator+=(Fp4& a, Fp4 b) { a = a + b; return a; }
ZKH_HD Fp4& operator*=(Fp4& a, Fp4 b) { a = a * b; return a; }
ZKH_HD Fp4 fp4_pow(Fp4 a, uint64_t e) {
    Fp4 r = Fp4::one();
    while (e) { if (e & 1) r = r * a; a = a * a; e >>= 1; }
    return r;
}
// Inverse through the norm tower: a(x) a(-x) = b0 + b2 x^2, (b0 + b2 x^2)(b0 - b2 x^2) = b0^2 + 11 b2^2 in Fp.
ZKH_HD Fp4 fp4_inv(Fp4 a) {
    const Fp beta = Fp::raw(BETA_M);
    Fp a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
    Fp b0 = a0 * a0 + beta * ((a1 + a1) * a3 - a2 * a2);
    Fp b2 = (a0 + a0) * a2 - a1 * a1 + beta * (a3 * a3);
    Fp ic = fp_inv(b0 * b0 + beta * (b2 * b2));
    b0 = b0 * ic; b2 = b2 * ic;
    return Fp4(a0, -a1, a2, -a3) * Fp4(b0, Fp::zero(), -b2, Fp::zero());
}

ZKH_HD uint32_t bitrev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
#endif
}
ZKH_HD uint32_t log2_ceil(uint64_t x) { uint32_t r = 0; while ((1ull << r) < x) r++; return r; }

}  // namespace zkh
