// common.hpp -- shared device helpers, record layout and host handle of the HIP env (see dcmrta_env.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/dcmrta_env.h"

namespace dcm {

constexpr int WAVE = 64;
constexpr int M = DCM_MAX_MEMBERS;
constexpr int AW_MAX = (DCM_MAX_AGENTS + 63) / 64;  // 64-bit words of an agent bitmask

// ---------------------------------------------------------------------------------- record
struct Hdr {  // 64 B header of an env record
    double now;            // current_time, env/task_env.py:28
    uint64_t seed;         // choice-protocol seed of this env
    uint64_t d;            // running decision counter (key of the choice protocol)
    double depot_x, depot_y;  // depot['location'] :111
    uint32_t flags;        // DCM_FLAG_*
    uint32_t groups;       // packed in memory: bits 0-7 cur_group, 8-15 n_groups, 16-23 empty_passes (see HdrRegs)
    uint32_t episodes;     // finished episodes since dcm_reset
    uint32_t reserved;
    double max_arrival;    // largest arrival time any agent_step of this episode has appended to an arrival_time list: the
                           // "max(arrival_time)" over all agents that next_decision / check_finished return as the time when
                           // nobody can decide any more (env/task_env.py:286,369).  With valid actions every list is
                           // monotone and this equals the maximum of the agents' last arrivals; a host-supplied action on a
                           // masked task can release an agent before it arrives, after which its list is not monotone
};
static_assert(sizeof(Hdr) == 64, "header must be 64 bytes");

// ainfo[a]: bit0 returned, bit1 assigned, bit2 in depot['members'], bit3 listed in members of route[-1],
//           bits 8-15 pending group id, bits 16-31 number of times moved to an abandoned_agent list
constexpr uint32_t A_RETURNED = 1u, A_ASSIGNED = 2u, A_INDEPOT = 4u, A_MEMBER = 8u, A_GRP = 0xFF00u;
// tinfo[t]: bits 0-7 requirements, 8-15 status (int8, may be stale: quirk Q3), 16-23 len(members),
//           bit 24 feasible_assignment, bit 25 finished
constexpr uint32_t T_FEAS = 1u << 24, T_FIN = 1u << 25;

__host__ __device__ constexpr uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }
// Abandonment log: for every agent the task ids whose abandoned_agent list it was appended to (env/task_env.py:265,271),
// in event order, up to AB_CAP per episode.  Lives in an HBM side table (touched only by the rare removal path and by the
// terminal metrics), so calculate_waiting_time's per-agent sums (:358-364) can be accumulated in the reference's order.
constexpr int AB_CAP = 16;
// Agents that overflow the log are summed from a dense count table instead: u16[A][T] per env (same side allocation, after
// the logs), cnt[a][t] = number of times task t moved agent a to its abandoned_agent list in this episode BEYOND the agent's
// first AB_CAP abandonments (which are in the log).  Incremented with a no-return 32-bit atomic on the containing word; never
// touched -- not even cleared -- while no agent overflows, i.e. at the reference's own constants.
__host__ __device__ constexpr size_t abcnt_pitch(int A, int T) { return (size_t)align16((uint32_t)(2 * A * T)); }
__host__ __device__ constexpr size_t side_bytes(int B, int A, int T) {
    return (size_t)B * A * AB_CAP * sizeof(uint16_t) + (size_t)B * abcnt_pitch(A, T);
}
// Record layout as a function of (A,T); see DESIGN.md §3.  All sections 8-byte aligned.
struct Lay {
    int A, T;
    int C = M;   // member slots per task: DCM_MAX_MEMBERS, or DCM_MAX_MEMBERS_WIDE for a DCM_PARAM_WIDE_MEMBERS handle
    __host__ __device__ constexpr uint32_t ax() const { return 64; }                 // f64[A] location x
    __host__ __device__ constexpr uint32_t ay() const { return 64 + 8 * A; }         // f64[A] location y
    __host__ __device__ constexpr uint32_t arr() const { return 64 + 16 * A; }       // f64[A] arrival_time[-1]
    __host__ __device__ constexpr uint32_t nd() const { return 64 + 24 * A; }        // f64[A] next_decision
    __host__ __device__ constexpr uint32_t tdist() const { return 64 + 32 * A; }     // f64[A] travel_dist
    __host__ __device__ constexpr uint32_t cur() const { return 64 + 40 * A; }       // i32[A] route[-1]
    __host__ __device__ constexpr uint32_t ainfo() const { return 64 + 44 * A; }     // u32[A]
    __host__ __device__ constexpr uint32_t tb() const { return 64 + 48 * A; }
    __host__ __device__ constexpr uint32_t ts() const { return tb(); }               // f64[T] time_start
    __host__ __device__ constexpr uint32_t tf() const { return tb() + 8 * T; }       // f64[T] time_finish
    __host__ __device__ constexpr uint32_t marr() const { return tb() + 16 * T; }    // f64[C][T] member arrivals
    __host__ __device__ constexpr uint32_t idw() const { return (uint32_t)(C + 7) / 8; }   // 64-bit id words per task
    __host__ __device__ constexpr uint32_t mids() const { return marr() + 8 * C * T; }   // u64[idw][T] ordered member ids, one byte each
    __host__ __device__ constexpr uint32_t tinfo() const { return mids() + 8 * idw() * T; }  // u32[T]
    __host__ __device__ constexpr uint32_t tnab() const { return tinfo() + 4 * T; }  // u32[T] len(abandoned_agent)
    __host__ __device__ constexpr uint32_t mut_bytes() const { return align16(tnab() + 4 * T); }
    __host__ __device__ constexpr uint32_t tx() const { return mut_bytes(); }        // f64[T] task x (instance)
    __host__ __device__ constexpr uint32_t ty() const { return mut_bytes() + 8 * T; }
    __host__ __device__ constexpr uint32_t tdur() const { return mut_bytes() + 16 * T; }
    __host__ __device__ constexpr uint32_t rec_bytes() const { return align16(mut_bytes() + 24 * T); }
    // LDS image of a record = the record + 48 B: abandonment-log pointer, incremental-update state, count-table pointer,
    // dirty-section mask, return-log pointer + capacity
    __host__ __device__ constexpr uint32_t aux() const { return rec_bytes(); }
    __host__ __device__ constexpr uint32_t lds_rec() const { return rec_bytes() + 48; }
    // Scratch of the terminal metrics (calculate_waiting_time), offsets relative to its own base: behind the record in LDS
    // for the one-chunk persistent and lockstep kernels (4 waves per SIMD by their VGPRs: up to 10 KB of LDS per env is free), a
    // per-env HBM buffer otherwise (an episode ends once in ~120 decisions; in LDS it would cost the general k_step, 7 waves per
    // SIMD, a quarter of its resident workgroups)
    __host__ __device__ constexpr uint32_t s_tw() const { return 0; }                                    // f64[T]
    __host__ __device__ constexpr uint32_t s_aw() const { return 8 * T; }                                // f64[A]
    __host__ __device__ constexpr uint32_t s_absort() const { return align16(8 * T + 8 * A); }           // u16[A][AB_CAP]
    // f64[A][list_cap]: every agent's member terms in ascending task order (kernels that gather them, see Sim::compute_waits), or
    // f64[T]: every task's latest arrival (the others); the replay kernels keep u32[2 T] here
    __host__ __device__ constexpr uint32_t s_terms() const { return align16(s_absort() + 2 * AB_CAP * A); }
    // entries of an agent's list (the tasks that list it at the end of an episode: ~ T x 3 / A on average), beyond which the
    // per-agent pass walks the tasks the slow way.  14 keeps the scratch of the one-chunk training shape (20A/50T) small enough for
    // 16 workgroups per CU; layouts with more than one task chunk never have their scratch in LDS and keep no lists
    __host__ __device__ constexpr uint32_t list_cap() const { return T <= 64 ? 14u : 0u; }
    __host__ __device__ constexpr uint32_t twords() const { return (uint32_t)(T + 63) / 64; }
    __host__ __device__ constexpr uint32_t s_amask() const {                                            // u64[A][twords]
        return s_terms() + (8u * (uint32_t)A * list_cap() > 8u * (uint32_t)T ? 8u * (uint32_t)A * list_cap() : 8u * (uint32_t)T);
    }
    __host__ __device__ constexpr uint32_t scratch_bytes() const { return align16(s_amask() + 8 * A * twords()); }
    __host__ __device__ constexpr uint32_t lds_bytes() const { return lds_rec() + scratch_bytes(); }     // record + scratch in LDS
};
static_assert(Lay{20, 50}.rec_bytes() == 5824, "S(20,50) = 64 + 48A + 96T");
static_assert(Lay{20, 50}.mids() == Lay{20, 50}.tb() + 56 * 50 && Lay{20, 50}.mut_bytes() == align16(Lay{20, 50}.tb() + 72 * 50), "5 member slots: the canonical record");
constexpr int MW = DCM_MAX_MEMBERS_WIDE;
static_assert(MW <= 16, "member ids: one byte each of up to two 64-bit words");

// Ordered member ids of one task: byte j of the word array = members[j] (the order matters: quirk Q1).  One word for the 5 slots
// of an ordinary handle (everything below folds to the single-word code), two for the 16 of a wide one.
template <int IW>
struct IdW {
    uint64_t w[IW];
    __host__ __device__ __forceinline__ uint32_t byte(int j) const {
        uint64_t x = w[0] >> (8 * (j & 7));
#pragma unroll
        for (int i = 1; i < IW; i++) if ((j >> 3) == i) x = w[i] >> (8 * (j & 7));
        return (uint32_t)x & 0xFFu;
    }
    __host__ __device__ __forceinline__ void put(int pos, uint32_t id) {      // byte `pos` must be zero (bytes above n always are)
#pragma unroll
        for (int i = 0; i < IW; i++) if ((pos >> 3) == i) w[i] |= (uint64_t)id << (8 * (pos & 7));
    }
    // position of `id` among the first n bytes, -1 if it is not listed (SWAR zero-byte search per word)
    __host__ __device__ __forceinline__ int find(uint32_t id, int n) const {
        int pos = -1;
#pragma unroll
        for (int i = IW - 1; i >= 0; i--) {
            const int ni = n - 8 * i;                                             // bytes of this word that are listed
            const uint64_t x = w[i] ^ (0x0101010101010101ull * (uint64_t)id);
            uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;  // lowest set bit = first zero byte
            z &= (ni >= 8) ? ~0ull : (ni <= 0 ? 0ull : ((1ull << (8 * ni)) - 1ull));
            if (z) pos = 8 * i + ((__builtin_ffsll((long long)z) - 1) >> 3);
        }
        return pos;
    }
};

struct KP {
    double mwt;       // max_waiting_time
    double max_time;  // MAX_TIME
};

// optional route history of the lockstep API (agent['route'] / agent['arrival_time'], env/task_env.py:95-96,314,318)
struct RouteLog {
    int16_t* task;    // [B][A][cap]  task id, -1 = depot
    double* arrival;  // [B][A][cap]
    int32_t* len;     // [B][A]
    int32_t cap;
};

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// One wavefront == one workgroup: LDS instructions of a wave execute in program order, so cross-lane
// hand-offs through LDS only need the COMPILER not to reorder across the hand-off.  A wavefront-scope
// fence does that without the vmcnt(0)/lgkmcnt(0) drain a workgroup-scope __syncthreads() would insert
// (which made every phase wait for the observation stores to reach memory).
#define WSYNC()                                               \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// ---------------------------------------------------------------------------------- uniform helpers
// does a generic pointer point into LDS?  (the aperture test on its high half; false in the host pass, which never calls it)
__device__ __forceinline__ bool in_lds(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)p);
#else
    (void)p;
    return false;
#endif
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni(uint64_t v) {
    return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}
__device__ __forceinline__ double uni(double v) { return __longlong_as_double((long long)uni((uint64_t)__double_as_longlong(v))); }

// The hot header fields as the kernels carry them (wave-uniform, in SGPRs); depot / episodes / max_arrival stay in the LDS record
struct HdrRegs {
    double now;
    uint64_t seed, d;
    uint32_t flags;
    int32_t cur_group;     // 1-based index of the group now deciding (worker.py:52), 0 = none
    int32_t n_groups;      // groups of the current event (env/task_env.py:291-298)
    int32_t empty_passes;  // consecutive zero-decider events (guard)
};
__device__ __forceinline__ HdrRegs load_hdr(const unsigned char* p) {
    const Hdr* q = (const Hdr*)p;
    HdrRegs h;
    h.now = uni(q->now); h.seed = uni(q->seed); h.d = uni(q->d);
    h.flags = uni(q->flags);
    const uint32_t g = uni(q->groups);
    h.cur_group = (int32_t)(g & 0xFFu); h.n_groups = (int32_t)((g >> 8) & 0xFFu); h.empty_passes = (int32_t)((g >> 16) & 0xFFu);
    return h;
}

// ---------------------------------------------------------------------------------- choice protocol
// dcmrta_amd/choice.py: stream of 32-bit words hi(key_1), lo(key_1), hi(key_2), ...;
// slot 0 leader, 1 action, 2+j follower j; below(r,n) = (r*n) >> 32.  All wave-uniform.
constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ULL;
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// (hipcc keeps the loop-carried decision counter in VGPRs, so this mix runs as ~25 VALU instructions; forcing it onto the
// scalar unit with v_readfirstlane was measured 1.1 % SLOWER -- the dependent s_mul chain lengthens the wave's critical path.)
__device__ __forceinline__ uint64_t key1(uint64_t seed, uint64_t d) { return mix64(seed + GAMMA * (d + 1)); }
__device__ __forceinline__ int below(uint32_t r, int n) { return (int)(((uint64_t)r * (uint64_t)(uint32_t)n) >> 32); }

// ---------------------------------------------------------------------------------- wave reductions (DPP)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);  // lanes without a source keep their own value
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// v_min_f64 / v_max_f64 return the other operand when one is a quiet NaN (IEEE minNum/maxNum) and are
// exact; written as asm so the compiler does not add a canonicalising v_max_f64 x,x in front of each.
__device__ __forceinline__ double nanmin2(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double nanmax2(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// NaN-ignoring minimum over the wave (np.nanmin), NaN when every lane is NaN.
__device__ __forceinline__ double wave_nanmin(double v) {
    v = nanmin2(v, dpp_f64<0x111, 0xF>(v));  // row_shr:1
    v = nanmin2(v, dpp_f64<0x112, 0xF>(v));  // row_shr:2
    v = nanmin2(v, dpp_f64<0x114, 0xF>(v));  // row_shr:4
    v = nanmin2(v, dpp_f64<0x118, 0xF>(v));  // row_shr:8  -> lane 15 of each row holds the row minimum
    v = nanmin2(v, dpp_f64<0x142, 0xA>(v));  // row_bcast:15 into rows 1,3
    v = nanmin2(v, dpp_f64<0x143, 0xC>(v));  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave minimum
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// The same when only lanes 0..NL-1 can hold a value (the others hold NaN): one or two reduction stages fewer
template <int NL>
__device__ __forceinline__ double wave_nanmin_n(double v) {
    if constexpr (NL > 32) return wave_nanmin(v);
    v = nanmin2(v, dpp_f64<0x111, 0xF>(v));
    v = nanmin2(v, dpp_f64<0x112, 0xF>(v));
    v = nanmin2(v, dpp_f64<0x114, 0xF>(v));
    v = nanmin2(v, dpp_f64<0x118, 0xF>(v));                    // lane 15 / 31: minimum of row 0 / 1
    if constexpr (NL > 16) v = nanmin2(v, dpp_f64<0x142, 0xA>(v));   // row_bcast:15 -> lane 31: minimum of rows 0 and 1
    constexpr int SRC = NL > 16 ? 31 : 15;
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC), hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_nanmax(double v) {
    v = nanmax2(v, dpp_f64<0x111, 0xF>(v));
    v = nanmax2(v, dpp_f64<0x112, 0xF>(v));
    v = nanmax2(v, dpp_f64<0x114, 0xF>(v));
    v = nanmax2(v, dpp_f64<0x118, 0xF>(v));
    v = nanmax2(v, dpp_f64<0x142, 0xA>(v));
    v = nanmax2(v, dpp_f64<0x143, 0xC>(v));
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// position of the idx-th (0-based) set bit of a wave-uniform mask, 0 <= idx < popcount(m); all 64 lanes must call.
// v_mbcnt_lo/hi gives every lane the number of set bits below it; the lanes up to and including the wanted bit are exactly those
// with at most idx of them, so one compare + one scalar popcount find it (an out-of-range idx gives -1 / 63: callers that probe
// several mask words discard those).
__device__ __forceinline__ int nth_set_bit(uint64_t m, int idx, int lane) {
    (void)lane;
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    return __popcll(__ballot(rank <= idx)) - 1;
}

// ---------------------------------------------------------------------------------- numpy add.reduce
// np.sum / np.mean use pairwise summation (8 accumulators per <=128-element block, recursive
// halving above); restated so the perf metrics of worker.py:103-108 are bit-identical.
__device__ __noinline__ double psum_block(const double* a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}
static_assert(DCM_MAX_AGENTS <= 128, "per-agent sums are taken as one pairwise block");
template <int DEPTH>
__device__ double psum(const double* a, int n) {
    if constexpr (DEPTH == 0) {
        return psum_block(a, n);
    } else {
        if (n <= 128) return psum_block(a, n);
        int n2 = n / 2;
        n2 -= n2 % 8;
        return psum<DEPTH - 1>(a, n2) + psum<DEPTH - 1>(a + n2, n - n2);
    }
}

// env/task_env.py:161-163 -- np.linalg.norm of a 2-vector == sqrt(fma(dy,dy,dx*dx)) on the
// reference machine (tests/golden/distance_kat.npz).
__device__ __forceinline__ double dist2(double ax, double ay, double bx, double by) {
    const double dx = ax - bx, dy = ay - by;
    return sqrt(__builtin_fma(dy, dy, dx * dx));
}
// travel_time = distance / velocity with velocity 0.2 (env/task_env.py:99,315).  The IEEE quotient d / 0.2 costs an
// 11-instruction v_div_scale / v_rcp_f64 / v_div_fmas sequence; Markstein's division by a constant gives the same
// correctly rounded result in three: 5.0 = RN(1/0.2) (1/0.2 = 5 - 2.8e-16, half an ulp below 5 is 4.4e-16),
// q = RN(5d) is a faithful quotient (|5d - d/0.2| = 2^-54 relative), r = d - 0.2 q is exact in one fma, and then
// RN(q + 5 r) = RN(d / 0.2) (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, thm. on
// "correcting a faithful quotient").  Needs no overflow/underflow in q and r: any distance between finite points
// that is 0 or in [2^-900, 2^1000].  Pinned against the true quotient by tests/test_gpu_parity.py (distance KATs,
// 2^22 random distances over 600 binades).
#ifndef DCM_IEEE_DIV
__device__ __forceinline__ double over_velocity(double d) {
    const double q = d * 5.0;
    const double r = __builtin_fma(-q, 0.2, d);
    return __builtin_fma(r, 5.0, q);
}
#else
__device__ __forceinline__ double over_velocity(double d) { return d / 0.2; }
#endif

// ---------------------------------------------------------------------------------- record I/O
__device__ __forceinline__ void copy16(unsigned char* dst, const unsigned char* src, uint32_t bytes, int lane) {
    const uint4* s = (const uint4*)src;
    uint4* d = (uint4*)dst;
    for (uint32_t i = lane; i < bytes / 16; i += WAVE) d[i] = s[i];
}
// record HBM -> LDS: streamed once, keep it out of the way of the observation stores in L2
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// LDS -> HBM with non-temporal stores: the lockstep kernels' write-back and observation rows at batches far larger than the L2s
// (nothing re-reads the lines before they would be evicted anyway; measured -5 % per 65 536-env launch, no change at 4096, where
// the next step's record loads still hit L2 and the plain stores stay)
__device__ __forceinline__ void copy16_nt(unsigned char* dst, const unsigned char* src, uint32_t bytes, int lane) {
    const u32x4* s = (const u32x4*)src;
    u32x4* d = (u32x4*)dst;
    for (uint32_t i = lane; i < bytes / 16; i += WAVE) __builtin_nontemporal_store(s[i], d + i);
}
__device__ __forceinline__ void copy16_in(unsigned char* dst, const unsigned char* src, uint32_t bytes, int lane) {
    const u32x4* s = (const u32x4*)src;
    u32x4* d = (u32x4*)dst;
    for (uint32_t i = lane; i < bytes / 16; i += WAVE) d[i] = __builtin_nontemporal_load(s + i);
}
// The same for a compile-time record size: ALL loads are issued before the first LDS write.  (hipcc compiles the loop above
// into load -> s_waitcnt vmcnt(0) -> ds_write per 1 KiB chunk, i.e. one full HBM round trip per chunk, six in a row for
// the 20A/50T record: 3.5 us of the lockstep kernel's 13 us wave lifetime in the round-3 phase profile.)
template <uint32_t BYTES, bool NT = true>
__device__ __forceinline__ void copy16_in_all(unsigned char* dst, const unsigned char* src, int lane) {
    constexpr uint32_t N16 = BYTES / 16, CH = (N16 + WAVE - 1) / WAVE, G = 8;   // groups of 8 chunks = 32 VGPRs in flight
    const u32x4* s = (const u32x4*)src;
    u32x4* d = (u32x4*)dst;
#pragma unroll
    for (uint32_t g = 0; g < CH; g += G) {
        u32x4 r[G];
        // no predication at all: the idle lanes of the partial last chunk re-copy the record's last 16 bytes (same value to
        // the same LDS address), so that the compiler keeps one straight run of global_load_dwordx4 followed by the writes
#pragma unroll
        for (uint32_t c = 0; c < G; c++) if (g + c < CH) {
            const uint32_t i = (g + c) * WAVE + lane;
            const u32x4* q = s + ((g + c + 1) * WAVE <= N16 ? i : (i < N16 ? i : N16 - 1));
            if constexpr (NT) r[c] = __builtin_nontemporal_load(q); else r[c] = *q;
        }
#pragma unroll
        for (uint32_t c = 0; c < G; c++) if (g + c < CH) {
            const uint32_t i = (g + c) * WAVE + lane;
            d[(g + c + 1) * WAVE <= N16 ? i : (i < N16 ? i : N16 - 1)] = r[c];
        }
    }
}
__device__ __forceinline__ void store_hdr(const HdrRegs& h, int lane) {
    if (lane == 0) {
        Hdr* q = (Hdr*)smem;
        q->now = h.now; q->seed = h.seed; q->d = h.d; q->flags = h.flags;
        q->groups = (uint32_t)h.cur_group | ((uint32_t)h.n_groups << 8) | ((uint32_t)h.empty_passes << 16);
    }
}

// ---------------------------------------------------------------------------------- host side
int fail(int code, const char* fmt, const char* a = "", const char* b = "");
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return dcm::fail(DCM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define CHECK_HANDLE(env) \
    if (!(env)) return dcm::fail(DCM_ERR_INVALID, "null env handle")
// every launching entry point runs on the CALLER's current device: refuse loudly when that is not the handle's device
// (device pointers of one GPU dereferenced by a kernel on another fault asynchronously and far from the cause)
#define CHECK_ENV(env)                                                                                        \
    do {                                                                                                      \
        if (!(env)) return dcm::fail(DCM_ERR_INVALID, "null env handle");                                     \
        int _cur = -1;                                                                                        \
        if (hipGetDevice(&_cur) != hipSuccess || _cur != (env)->p.device)                                     \
            return dcm::fail(DCM_ERR_STATE, "the current HIP device is not the device this env was created on"); \
    } while (0)
#define LAUNCH_OK() HIP_TRY(hipGetLastError())
#define GRID(env) dim3((env)->p.n_envs), dim3(dcm::WAVE)

}  // namespace dcm

struct dcm_env {
    dcm_params p;
    int A = 0, T = 0;                // batch dims = p.n_agents / p.n_tasks: shapes of every array crossing the ABI
    dcm::Lay L;                      // record layout dims (>= the batch dims: Lay{20,50} for every shape inside the reference's
                                     // training range, Lay{64,64} for the other one-chunk shapes, so that those shapes share
                                     // constant-offset kernel instantiations)
    dcm::KP kp;
    unsigned char* state = nullptr;  // [B][rec_bytes]
    unsigned char* gscratch = nullptr;  // [B][scratch_bytes] terminal-metrics scratch of the kernels that keep it out of LDS
    double* summary = nullptr;       // [B][8]
    uint16_t* ablog = nullptr;       // [B][A][AB_CAP] abandonment log (side table of the state)
    bool loaded = false, reset_done = false;
    int32_t* sizes = nullptr;        // [B][2] (A_e, T_e) of a ragged batch (dcm_load_instances_ragged), else nullptr
    std::vector<int32_t> sizes_host;
    // route replay (dcmrta_replay.hip)
    int32_t* routes = nullptr;       // [B][A][route_cap] actions
    int32_t* route_len = nullptr;    // [B][A], -1 = pre_set_route is None
    int32_t route_cap = 0, member_cap = 0;
    double* rmarr = nullptr;         // replay scratch (dcm_load_routes): per env member arrival times [T][member_cap], time_finish [T], travel_dist and max arrival [A]
    int32_t replay_placement = 0;    // dcm_set_replay_placement: 0 auto, 1 replay scratch in LDS, 2 in HBM
    int32_t vis[4] = {20, 20, 10, 100};  // dynamic-arrival schedule: initial, batch, period, cap (env/task_env.py:567,:221)
    dcm::RouteLog log{nullptr, nullptr, nullptr, 0};   // dcm_set_route_log
    double* retlog = nullptr;        // dcm_set_return_log: [B][retcap] ring of episode returns
    int32_t retcap = 0;
    // Deferred terminal metrics of the lockstep API (step_fast.hpp: dcm_step with DCM_PARAM_AUTO_RESET on a one-chunk layout): the wave
    // that ends an episode parks the final record here instead of running calculate_waiting_time itself -- it is the slowest wave of
    // its launch -- and k_terminal_flush computes reward + metrics from the snapshots later (every FLUSH_EVERY steps, and before
    // anything reads or writes the summary rows)
    unsigned char* side = nullptr;   // [B][side_pitch]: record image + the env's abandonment rows at the end of the episode
    uint32_t* pendq = nullptr;       // [B]: 1 = the env's snapshot is waiting for k_terminal_flush
    uint32_t side_pitch = 0;
    // ... and restarts from a copy of the records dcm_reset produced (reset_state + the first event depend on the instance only)
    // instead of recomputing them; dropped by anything that changes instances or records behind the kernel's back
    unsigned char* init = nullptr;   // [B][rec_bytes]
    bool init_valid = false, init_failed = false;
    int steps_since_flush = 0;
    bool maybe_pending = false, side_failed = false;
    bool captured = false;           // a dcm_step of this handle was captured into a graph: every later step computes its metrics inline
    static constexpr int FLUSH_EVERY = 32;
};

namespace dcm {
// compute the summary rows of the snapshots that are waiting (no-op when none can be); every entry point that reads or writes
// summary rows calls it first
int flush_pending(dcm_env* env, void* stream);
// forget them (the envs are being reset / reloaded)
int drop_pending(dcm_env* env, void* stream);
}  // namespace dcm
