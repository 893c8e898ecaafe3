// qe_device.h -- device primitives of the gfx950 Q-learning engine: counter-based draws, the
// lane-group view of a Q-table row, arg-max/tie selection, TD arithmetic.
//
// Row view: a row of `A` Q-values (stride `ld`, multiple of 4) is spread over a group of L
// consecutive lanes (L = power of two, 1..64, L*4 >= ld); lane `sub` holds columns 4*sub..4*sub+3
// from one 16-byte (fp32) / two 16-byte (fp64) loads.  A 64-wide wavefront therefore handles 64/L
// agents; reductions stay inside the group: DPP moves when the width is a compile-time constant of
// 2/4/8/16 lanes, `__shfl_*` (ds_bpermute) otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qe {

constexpr uint32_t STREAM_POLICY = 0;
constexpr uint32_t STREAM_ENV = 1;

struct U4 {
    uint32_t x, y, z, w;
};

// Philox4x32-10 (Salmon et al. SC'11); same constants as oracle/draws.py.
__host__ __device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                     uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {  // murmur3 fmix32
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * b) >> 32);
}

// ---------------------------------------------------------------------------------------------
template <typename T>
struct Row4 {
    T v[4];
};

template <typename T>
__device__ __forceinline__ T neg_inf() {
    return -__builtin_huge_val();
}
template <>
__device__ __forceinline__ float neg_inf<float>() {
    return -__builtin_huge_valf();
}

// Plain (cached) row load: used where no other workgroup writes the row inside this launch.
__device__ __forceinline__ Row4<float> load_row4(const float* q, int64_t row, int ld, int sub) {
    Row4<float> r;
    const int c = 4 * sub;
    if (c < ld) {
        const float4 f = *reinterpret_cast<const float4*>(q + row * ld + c);
        r.v[0] = f.x; r.v[1] = f.y; r.v[2] = f.z; r.v[3] = f.w;
    } else {
        r.v[0] = r.v[1] = r.v[2] = r.v[3] = neg_inf<float>();
    }
    return r;
}
__device__ __forceinline__ Row4<double> load_row4(const double* q, int64_t row, int ld, int sub) {
    Row4<double> r;
    const int c = 4 * sub;
    if (c < ld) {
        const double2* p = reinterpret_cast<const double2*>(q + row * ld + c);
        const double2 a = p[0], b = p[1];
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    } else {
        r.v[0] = r.v[1] = r.v[2] = r.v[3] = neg_inf<double>();
    }
    return r;
}

// Coherent element access (agent scope, bypasses the CU's L1): used by the ordered path, where
// other waves of the same launch have written the table.
template <typename T>
__device__ __forceinline__ T load_live(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void store_live(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ Row4<T> load_row4_live(const T* q, int64_t row, int ld, int sub) {
    Row4<T> r;
    const int c = 4 * sub;
    if (c < ld) {
        const T* p = q + row * ld + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.v[j] = load_live(p + j);
    } else {
        r.v[0] = r.v[1] = r.v[2] = r.v[3] = neg_inf<T>();
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// Lane-group data exchange.  hipcc lowers every __shfl* to ds_bpermute_b32 -- an LDS-crossbar round
// trip (address arithmetic + ~100 cycles when the result is needed at once), and the selection of one
// action is a chain of a dozen dependent ones.  When the group width is a compile-time constant of 2
// or 4 lanes (`LC`; groups are quad-aligned) the same exchanges are single DPP moves (quad_perm /
// row_shr), issued by the VALU at full rate.  Partners and operation order are those of the generic
// code, so results are identical bit for bit.  LC == 0 (runtime width `L`) and wider groups keep the
// generic shuffles.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) { return (unsigned)dpp_mov<CTRL>((int)v); }
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) { return __int_as_float(dpp_mov<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    return __hiloint2double(dpp_mov<CTRL>(__double2hiint(v)), dpp_mov<CTRL>(__double2loint(v)));
}
constexpr int DPP_XOR1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int DPP_SHR1 = 0x111;    // row_shr:1
constexpr int DPP_SHR2 = 0x112;    // row_shr:2
constexpr int DPP_SHR4 = 0x114;    // row_shr:4
constexpr int DPP_SHR8 = 0x118;    // row_shr:8
constexpr int DPP_HALF_MIRROR = 0x141;  // lane k <-> 7-k inside each 8 lanes
constexpr int DPP_MIRROR = 0x140;       // lane k <-> 15-k inside each row of 16 lanes
constexpr bool dpp_width(int LC) { return LC == 2 || LC == 4 || LC == 8 || LC == 16; }
// Reductions over 8 / 16 lanes pair every lane with its mirror image first (there is no xor-4 / xor-8
// DPP pattern): after mirror and half-mirror, lane j of a quad holds the combination of lanes
// {j, 7-j, 8+j, 15-j}, and the two quad steps complete the group.

// value held by lane K of the group (K a compile-time constant)
template <int LC, int K, typename T>
__device__ __forceinline__ T group_bcast(T v, int L) {
    if constexpr (LC == 4) return dpp_mov<(K & 3) * 0x55>(v);           // quad_perm:[K,K,K,K]
    else if constexpr (LC == 2) return dpp_mov<K ? 0xF5 : 0xA0>(v);     // quad_perm:[K,K,2+K,2+K]
    else return __shfl(v, K, LC ? LC : L);
}

template <int LC = 0, typename T>
__device__ __forceinline__ T group_max(T v, int L) {
    if constexpr (LC == 16) {
        const T o = dpp_mov<DPP_MIRROR>(v);
        v = o > v ? o : v;
    }
    if constexpr (LC >= 8 && dpp_width(LC)) {
        const T o = dpp_mov<DPP_HALF_MIRROR>(v);
        v = o > v ? o : v;
    }
    if constexpr (LC >= 4 && dpp_width(LC)) {
        const T o = dpp_mov<DPP_XOR2>(v);
        v = o > v ? o : v;
    }
    if constexpr (dpp_width(LC)) {
        const T o = dpp_mov<DPP_XOR1>(v);
        return o > v ? o : v;
    } else {
        if (LC) L = LC;
#pragma unroll
        for (int off = L >> 1; off > 0; off >>= 1) {
            const T o = __shfl_xor(v, off, L);
            v = o > v ? o : v;
        }
        return v;
    }
}
template <int LC = 0>
__device__ __forceinline__ int group_max_int(int v, int L) {
    return group_max<LC, int>(v, L);
}
// bitwise OR over the group (used to fetch the value of ONE lane chosen at run time: the others
// contribute zero)
template <int LC>
__device__ __forceinline__ unsigned group_or(unsigned v) {
    static_assert(dpp_width(LC), "DPP widths only");
    if constexpr (LC == 16) v |= dpp_mov<DPP_MIRROR>(v);
    if constexpr (LC >= 8) v |= dpp_mov<DPP_HALF_MIRROR>(v);
    if constexpr (LC >= 4) v |= dpp_mov<DPP_XOR2>(v);
    return v | dpp_mov<DPP_XOR1>(v);
}
template <int LC>
__device__ __forceinline__ float group_pick(float mine, bool chosen) {
    return __uint_as_float(group_or<LC>(chosen ? __float_as_uint(mine) : 0u));
}
template <int LC>
__device__ __forceinline__ double group_pick(double mine, bool chosen) {
    const unsigned hi = group_or<LC>(chosen ? (unsigned)__double2hiint(mine) : 0u);
    const unsigned lo = group_or<LC>(chosen ? (unsigned)__double2loint(mine) : 0u);
    return __hiloint2double((int)hi, (int)lo);
}

// max over the valid columns of this lane's 4 elements, reduced over the group (-inf if none).  A NaN column is
// SKIPPED: the scan of the reference's list variants (`if v > max_val`, q_learning_optimal.py:290-296, :337-344).
template <int LC = 0, typename T>
__device__ __forceinline__ T row_max_skipnan(const Row4<T>& row, uint32_t valid4, int L) {
    T m = neg_inf<T>();
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((valid4 >> j) & 1u) m = row.v[j] > m ? row.v[j] : m;
    return group_max<LC>(m, L);
}

template <typename T>
__device__ __forceinline__ T quiet_nan() {
    return (T)__builtin_nanf("");
}

// whether some valid column of the group's row holds a NaN (all lanes of a lane group call it together; groups
// are aligned to their width): one ballot, each group looks at its own bits.
template <int LC = 0, typename T>
__device__ __forceinline__ bool row_has_nan(const Row4<T>& row, uint32_t valid4, int L) {
    if (LC) L = LC;
    bool mine = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) mine |= ((valid4 >> j) & 1u) && row.v[j] != row.v[j];
    if constexpr (dpp_width(LC)) return group_or<LC>(mine ? 1u : 0u) != 0u;  // (compile-time widths: a few DPP moves)
    const unsigned long long b = __ballot(mine);
    if (L >= 64) return b != 0ull;
    const int base = (int)__lane_id() & ~(L - 1);
    return ((b >> base) & ((1ull << L) - 1ull)) != 0ull;
}

// np.max over the valid columns (q_learning_optimal.py:548, :757-761, :884-888; the TD target of every learn variant
// and the greedy pick of the NumPy selection variants): NaN as soon as one valid column holds a NaN.
template <int LC = 0, typename T>
__device__ __forceinline__ T row_max_valid(const Row4<T>& row, uint32_t valid4, int L) {
    const T m = row_max_skipnan<LC>(row, valid4, L);
    return row_has_nan<LC>(row, valid4, L) ? quiet_nan<T>() : m;
}

// columns < A of this lane, as a 4-bit field
__device__ __forceinline__ uint32_t in_range4(int sub, int A) {
    const int rem = A - 4 * sub;
    return rem >= 4 ? 0xFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
}

// Epsilon-greedy pick (all reference choose_action* variants share this distribution,
// q_learning_optimal.py:263-642): explore -> k-th valid action, k = mulhi(x1, n_valid);
// greedy -> k-th action tied at the valid maximum, k = mulhi(x2, n_ties).  Returns -1 when no
// action is selectable (-2: because the maximum is NaN).  *picked_q receives Q[s, action] as held in `row`.
// `nan_max`: the row maximum is NumPy's (NaN when a valid column holds one: nothing ties with it, no greedy pick is
// possible and the reference's random.choice raises, :430, :470, :563, :628) -- the NumPy variants; false = the
// list variants' scan, which steps over NaN columns (:290-296, :337-344).
template <int LC = 0, typename T>
__device__ __forceinline__ int select_action(const Row4<T>& row, uint32_t valid4, int sub, int L,
                                             bool explore, uint32_t x1, uint32_t x2, T* picked_q, bool nan_max) {
    if (LC) L = LC;
    T m = row_max_skipnan<LC>(row, valid4, L);
    if (nan_max && row_has_nan<LC>(row, valid4, L)) m = quiet_nan<T>();
    uint32_t f = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = ((valid4 >> j) & 1u) && (explore || row.v[j] == m);
        f |= (ok ? 1u : 0u) << j;
    }
    const int cnt = __popc(f);
    int incl = cnt;
    int total;
    if constexpr (dpp_width(LC)) {
        // row_shr reaches across the group boundary inside a row of 16 lanes; the `sub >=` guards
        // discard exactly those values
        int t = dpp_mov<DPP_SHR1>(incl);
        if (sub >= 1) incl += t;
        if constexpr (LC >= 4) {
            t = dpp_mov<DPP_SHR2>(incl);
            if (sub >= 2) incl += t;
        }
        if constexpr (LC >= 8) {
            t = dpp_mov<DPP_SHR4>(incl);
            if (sub >= 4) incl += t;
        }
        if constexpr (LC == 16) {
            t = dpp_mov<DPP_SHR8>(incl);
            if (sub >= 8) incl += t;
        }
        // the inclusive counts never decrease along the group, so their maximum is the total
        if constexpr (LC <= 4) total = group_bcast<LC, LC - 1>(incl, L);
        else total = group_max<LC, int>(incl, L);
    } else {
#pragma unroll
        for (int off = 1; off < L; off <<= 1) {
            const int t = __shfl_up(incl, off, L);
            if (sub >= off) incl += t;
        }
        total = __shfl(incl, L - 1, L);
    }
    int act = -1;
    if (total > 0) {
        const int k = (int)mulhi32(explore ? x1 : x2, (uint32_t)total);
        const int excl = incl - cnt;
        if (k >= excl && k < incl) {
            uint32_t g = f;
            for (int r = k - excl; r > 0; --r) g &= g - 1u;
            act = 4 * sub + (__ffs(g) - 1);
        }
    }
    act = group_max_int<LC>(act, L);
    // (NaN maximum: "no candidate" of a different kind than an empty / all -inf candidate set, which the NumPy
    // variants tie everywhere -- callers that implement that quirk must tell the two apart)
    if (m != m && !explore) act = -2;
    const int jj = act & 3;
    const T mine = jj == 0 ? row.v[0] : (jj == 1 ? row.v[1] : (jj == 2 ? row.v[2] : row.v[3]));
    const int holder = act < 0 ? 0 : (act >> 2);
    if constexpr (dpp_width(LC)) *picked_q = group_pick<LC>(mine, sub == holder);
    else *picked_q = __shfl(mine, holder, L);
    return act;
}

// ---------------------------------------------------------------------------------------------
// TD arithmetic.  Compiled with -ffp-contract=off: every operation rounds exactly once, in the
// order the reference performs it, so fp32 results equal the reference run on a float32 table and
// fp64 results equal the reference's default float64 table, bit for bit.
//
// ITER = single_learn (q_learning_optimal.py:728-768) under NEP-50 promotion:
//        t = g*m ; y = r + t ; d = y - q ; u = lr*d ; q' = q + u          (all in the table dtype)
// VEC  = _learn_vec (:819-891): on a float32 table `(1 - terminated)` is int64, which promotes the
//        target to float64, and np.add.at (:249) adds the float64 increment to the float32 cell in
//        float64 before rounding once: q' = (float)((double)q + lr*(y - q)).
// `apply` returns the new cell value (exclusive writer); `delta` the increment handed to
// atomicAdd when several transitions of one batch collide on a cell (VEC only).
struct Hyper {
    double gamma, lr;
    float gamma32, lr32;
};

template <typename T>
struct Td;

template <>
struct Td<float> {
    static __device__ __forceinline__ double vec_inc(float q, float r, float m, bool term, const Hyper& h) {
        // learn_vec multiplies by (1 - terminated) instead of selecting (q_learning_optimal.py:889): a
        // terminated transition whose next row has an infinite maximum yields inf * 0 = NaN there too
        const float t32 = h.gamma32 * m;
        const double t = (double)t32 * (term ? 0.0 : 1.0);
        const double y = (double)r + t;
        const double d = y - (double)q;
        return h.lr * d;
    }
    static __device__ __forceinline__ float apply(float q, float r, float m, bool term,
                                                  const Hyper& h, int mode, float* inc) {
        if (mode == 0) {
            const float t = term ? 0.0f : h.gamma32 * m;
            const float y = r + t;
            const float d = y - q;
            const float u = h.lr32 * d;
            *inc = u;
            return q + u;
        }
        const double u = vec_inc(q, r, m, term, h);
        *inc = (float)u;
        return (float)((double)q + u);
    }
    static __device__ __forceinline__ float delta(float q, float r, float m, bool term, const Hyper& h) {
        return (float)vec_inc(q, r, m, term, h);
    }
};

template <>
struct Td<double> {
    // vec = learn_vec arithmetic: gamma * max * (1 - terminated), see Td<float>::vec_inc
    static __device__ __forceinline__ double delta(double q, float r, double m, bool term, const Hyper& h,
                                                   bool vec = false) {
        const double t = vec ? (h.gamma * m) * (term ? 0.0 : 1.0) : (term ? 0.0 : h.gamma * m);
        const double y = (double)r + t;
        const double d = y - q;
        return h.lr * d;
    }
    static __device__ __forceinline__ double apply(double q, float r, double m, bool term,
                                                   const Hyper& h, int mode, double* inc) {
        const double u = delta(q, r, m, term, h, mode == 1);
        *inc = u;
        return q + u;
    }
};

}  // namespace qe
