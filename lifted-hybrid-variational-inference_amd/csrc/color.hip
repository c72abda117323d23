// color.hip -- colour refinement (colour passing) half-rounds on the GPU, integer only.
//
// Reference semantics: CompressedGraphWithObs.py:47-76 (SuperRV.split_by_structure),
// :152-175 (SuperF.split_by_structure); CompressedGraphSorted.py:40-61,90-116 use the same signatures.
//   factor signature = (old factor colour, tuple of its scope's rv colours, sorted iff potential.symmetric)
//   rv signature     = (old rv colour, sorted multiset of the colours of its incident factors)
// Only the induced *partition* matters (cluster objects have no stable ids in the reference), so a new
// colour is the dense rank of the signature.
//
// Exactness: signatures are reduced to two independent 64-bit fingerprints (h1, h2).  Items are ranked by
// h1 (radix sort); if two adjacent items agree on h1 but not on h2 a collision flag is raised and the
// host retries with another seed, so a wrong merge needs a simultaneous 128-bit collision.  The rv
// fingerprint is a *sum* of per-neighbour mixes: integer addition commutes, so the multiset needs no
// sorting and the atomics below are deterministic.
//
// Two ways to turn fingerprints into dense colours, with identical results (colour = rank of h1 among the distinct h1):
//   method 0  hash table: the kernel that builds the fingerprints also inserts each wavefront's distinct keys in an
//             open-addressing table of TABLE_SLOTS 64-bit keys (the answer has few distinct keys: ~30 k at 10 M edges), only the
//             distinct keys are ranked (bucket counting, no general sort), and every item looks its colour up.  More than
//             TABLE_SLOTS / 2 distinct keys raise the overflow flag: the caller repeats the half round with method 1.
//   method 1  radix sort of all n keys + flag + scan + scatter (n log n traffic: ~120 B per edge per round).
// Bound: HBM (the colour gathers of the signature kernels).
#include "common.hpp"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace lhvi {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

constexpr uint64_t SEED1 = 0x243F6A8885A308D3ull, SEED2 = 0x13198A2E03707344ull;

constexpr uint64_t EMPTY_KEY = ~0ull;          // never a fingerprint: h1 == EMPTY_KEY is mapped to EMPTY_KEY - 1 in both methods
__device__ __forceinline__ uint64_t key_of(uint64_t h) { return h == EMPTY_KEY ? EMPTY_KEY - 1 : h; }

__device__ __forceinline__ void factor_sig(const lhvi_graph_t& g, const uint8_t* __restrict__ symmetric,
                                           const int32_t* __restrict__ rv_color, const int32_t* __restrict__ f_color,
                                           uint64_t seed, int f, uint64_t& o1, uint64_t& o2) {
    const int base = g.fac_ptr[f];
    const int a = min(g.fac_ptr[f + 1] - base, LHVI_MAX_ARITY);
    int32_t c[LHVI_MAX_ARITY];
#pragma unroll
    for (int p = 0; p < LHVI_MAX_ARITY; ++p) c[p] = p < a ? rv_color[g.edge_var[base + p]] : -1;
    if (symmetric && symmetric[f]) {            // tuple(sorted(...)) for symmetric potentials
#pragma unroll
        for (int i = 1; i < LHVI_MAX_ARITY; ++i)
#pragma unroll
            for (int j = LHVI_MAX_ARITY - 1; j >= i; --j)
                if (j < a && c[j] < c[j - 1]) { int32_t t = c[j]; c[j] = c[j - 1]; c[j - 1] = t; }
    }
    uint64_t a1 = mix64((uint64_t)(uint32_t)f_color[f] ^ seed ^ SEED1);
    uint64_t a2 = mix64((uint64_t)(uint32_t)f_color[f] * 0x100000001B3ull + seed + SEED2);
#pragma unroll
    for (int p = 0; p < LHVI_MAX_ARITY; ++p) {
        if (p < a) {
            a1 = mix64(a1 * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)c[p] + 1);
            a2 = mix64(a2 ^ (((uint64_t)(uint32_t)c[p] + 0x51ull) * 0xD6E8FEB86659FD93ull));
        }
    }
    o1 = key_of(a1); o2 = a2;
}

__global__ void __launch_bounds__(BLOCK) factor_sig_kernel(lhvi_graph_t g, const uint8_t* __restrict__ symmetric,
                                                          const int32_t* __restrict__ rv_color,
                                                          const int32_t* __restrict__ f_color, uint64_t seed,
                                                          uint64_t* __restrict__ h1, uint64_t* __restrict__ h2,
                                                          uint32_t* __restrict__ idx) {
    int f = blockIdx.x * BLOCK + threadIdx.x;
    if (f >= g.F) return;
    uint64_t a1, a2;
    factor_sig(g, symmetric, rv_color, f_color, seed, f, a1, a2);
    h1[f] = a1; h2[f] = a2; idx[f] = (uint32_t)f;
}

__global__ void __launch_bounds__(BLOCK) rv_sig_init_kernel(int V, const int32_t* __restrict__ rv_color, uint64_t seed,
                                                           uint64_t* __restrict__ h1, uint64_t* __restrict__ h2,
                                                           uint32_t* __restrict__ idx) {
    int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= V) return;
    h1[v] = mix64((uint64_t)(uint32_t)rv_color[v] ^ seed ^ SEED2);
    h2[v] = mix64(((uint64_t)(uint32_t)rv_color[v] + seed) * 0xA24BAED4963EE407ull + SEED1);
    idx[v] = (uint32_t)v;
}

// Add the incident factors' colour mixes into the variable's two fingerprints.  Integer sums commute, so any order
// gives the same bits: a thread walks the CSR row of a variable with up to HUB_DEGREE edges, a wavefront shares the row
// of a hub (template variables of relational models have thousands of edges).  No atomics: 20 M 64-bit atomic adds on
// a 10 M-edge graph cost 3 ms per round, the segmented sums 0.3.
constexpr int HUB_DEGREE = LHVI_HUB_DEGREE;

__device__ __forceinline__ void sig_terms(const lhvi_graph_t& g, const int32_t* __restrict__ f_color, uint64_t seed, int k,
                                          uint64_t& a, uint64_t& b) {
    const uint64_t c = (uint64_t)(uint32_t)f_color[g.edge_fac[g.var_edge[k]]];
    a += mix64(c ^ seed ^ SEED1);
    b += mix64((c + 0x7Full) * 0xC2B2AE3D27D4EB4Full + seed);
}

__global__ void __launch_bounds__(BLOCK) rv_sig_accum_kernel(lhvi_graph_t g, const int32_t* __restrict__ f_color,
                                                            uint64_t seed, uint64_t* __restrict__ h1,
                                                            uint64_t* __restrict__ h2) {
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= g.V) return;
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo > HUB_DEGREE) return;
    uint64_t a = 0, b = 0;
    for (int k = lo; k < hi; ++k) sig_terms(g, f_color, seed, k, a, b);
    h1[v] = key_of(h1[v] + a); h2[v] += b;
}

__global__ void __launch_bounds__(BLOCK) rv_sig_accum_hub_kernel(lhvi_graph_t g, const int32_t* __restrict__ f_color,
                                                                uint64_t seed, uint64_t* __restrict__ h1,
                                                                uint64_t* __restrict__ h2) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= (g.hub_vars ? g.n_hubs : g.V)) return;
    const int v = g.hub_vars ? g.hub_vars[i] : i;          // without a hub list: scan every variable
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo <= HUB_DEGREE) return;
    uint64_t a = 0, b = 0;
    for (int k = lo + lane; k < hi; k += 64) sig_terms(g, f_color, seed, k, a, b);
    for (int off = 32; off > 0; off >>= 1) {
        a += ((uint64_t)(uint32_t)__shfl_xor((int)(a >> 32), off) << 32) | (uint32_t)__shfl_xor((int)a, off);
        b += ((uint64_t)(uint32_t)__shfl_xor((int)(b >> 32), off) << 32) | (uint32_t)__shfl_xor((int)b, off);
    }
    if (lane == 0) { h1[v] = key_of(h1[v] + a); h2[v] += b; }
}

// ---- method 0: hash table -------------------------------------------------------------------------------------------------
// Per half round: (1) the signature kernels store every item's fingerprints and insert the DISTINCT keys of each wavefront
// (neighbouring items of a relational graph mostly share their signature) with compare-and-swap -- no plain load ever probes
// the table inside the kernel that fills it: per-XCD L2s are not coherent, a slot once read as empty would stay empty for that
// XCD and every later item would fall back to the atomic; (2) the distinct keys are ranked without a general sort: counted
// into 8 192 buckets by their top 13 bits while they are inserted, one block scans the counts, the keys are scattered into
// their bucket's segment and each key counts the smaller keys of its segment (a handful) -- rank = numeric order of the
// 64-bit keys, exactly what the radix sort of method 1 gives; (3) every item finds its key with plain loads (a new kernel:
// the table is now read-only) and takes the slot's rank as its colour.
constexpr int TABLE_BITS = 20;
constexpr uint32_t TABLE_SLOTS = 1u << TABLE_BITS;          // 1 M slots: up to 512 k distinct colours per half round
constexpr uint32_t MAX_DISTINCT = TABLE_SLOTS / 2;
constexpr int MAX_PROBES = 8192;
constexpr int BUCKET_BITS = 13;
constexpr int BUCKETS = 1 << BUCKET_BITS;        // by the keys' top bits: ~4 keys per bucket at 30 k colours, 64 at the table's limit

struct Table {
    uint64_t* keys;         // [TABLE_SLOTS] EMPTY_KEY or a fingerprint h1
    uint64_t* h2;           // [TABLE_SLOTS] h2 of the item that inserted the key
    int32_t* rank;          // [TABLE_SLOTS] colour of the slot's key
    uint64_t* dkeys;        // [MAX_DISTINCT] distinct keys in insertion order
    uint32_t* dslot;        // [MAX_DISTINCT] their slots
    uint64_t* skeys;        // [MAX_DISTINCT] the same, grouped by bucket
    uint32_t* sslot;
    uint32_t* bucket;       // [BUCKETS] keys per bucket | [BUCKETS + 1] segment starts | [BUCKETS] scatter cursors
    uint32_t* count;        // [2] number of distinct keys, overflow flag
};
__device__ __forceinline__ uint32_t* bucket_count(const Table& t) { return t.bucket; }
__device__ __forceinline__ uint32_t* bucket_start(const Table& t) { return t.bucket + BUCKETS; }
__device__ __forceinline__ uint32_t* bucket_cursor(const Table& t) { return t.bucket + 2 * BUCKETS + 1; }
__device__ __forceinline__ uint32_t home_slot(uint64_t key) { return (uint32_t)(key >> 17) & (TABLE_SLOTS - 1); }

// make sure `key` is in the table.  The probe reads slots with agent-scope (sc1) loads: per-XCD L2s are not coherent, and a
// slot a plain load once showed empty would stay empty for that XCD, sending every later item of the key to the atomic;
// slots only ever go from EMPTY_KEY to a key, so whatever key a load shows is final, and only an empty slot is settled by
// the compare-and-swap.
__device__ __forceinline__ void insert_key(const Table& t, uint64_t key, uint64_t h2) {
    uint32_t s = home_slot(key);
    for (int probe = 0; probe < MAX_PROBES; ++probe) {
        uint64_t cur = __hip_atomic_load(&t.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == EMPTY_KEY) {
            cur = atomicCAS((unsigned long long*)&t.keys[s], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (cur == EMPTY_KEY) {                     // this lane inserted the key: it owns the slot's h2 and the list entry
                t.h2[s] = h2;
                const uint32_t pos = atomicAdd(&t.count[0], 1u);
                if (pos < MAX_DISTINCT) {
                    t.dkeys[pos] = key; t.dslot[pos] = s;
                    atomicAdd(&bucket_count(t)[key >> (64 - BUCKET_BITS)], 1u);
                } else t.count[1] = 1;
                return;
            }
        }
        if (cur == key) return;
        s = (s + 1) & (TABLE_SLOTS - 1);
    }
    t.count[1] = 1;                                     // a probe sequence this long means the table is (nearly) full
}

// Neighbouring items of a relational graph mostly share their signature, so a workgroup filters its keys through a small
// direct-mapped LDS table before going to memory: a lane publishes (key with its low 8 bits replaced by its thread id) in
// the key's LDS slot -- one 8-byte store, the last writer wins whole -- and reads the slot back: the winner inserts the key
// for everyone, lanes that find another thread's entry for the same key are done, lanes that lost the slot to a different key
// insert theirs themselves.  Later waves of the workgroup find the entry and skip without writing.  (Keys equal in all but
// their low 8 bits would share an entry, 2^-56 per pair; the key then missing from the table shows up in assign_kernel as
// an empty slot, which is reported as overflow and sends the half round through the sort.)
constexpr int FILTER_SLOTS = 512;
__device__ __forceinline__ void block_insert(const Table& t, uint64_t* __restrict__ filter, uint64_t key, uint64_t h2, bool active) {
    const uint64_t tag = key & ~0xFFull, mine = tag | threadIdx.x;
    uint64_t* slot = filter + ((key >> 9) & (FILTER_SLOTS - 1));
    bool go = active && (*slot & ~0xFFull) != tag;
    if (go) *slot = mine;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (go) {
        const uint64_t now = *slot;
        go = now == mine || (now & ~0xFFull) != tag;
    }
    if (go) insert_key(t, key, h2);
}

__global__ void __launch_bounds__(BLOCK) factor_insert_kernel(lhvi_graph_t g, const uint8_t* __restrict__ symmetric,
                                                             const int32_t* __restrict__ rv_color,
                                                             const int32_t* __restrict__ f_color, uint64_t seed,
                                                             uint64_t* __restrict__ h1, uint64_t* __restrict__ h2, Table t) {
    __shared__ uint64_t filter[FILTER_SLOTS];
    for (int i = threadIdx.x; i < FILTER_SLOTS; i += BLOCK) filter[i] = EMPTY_KEY;
    __syncthreads();
    const int f = blockIdx.x * BLOCK + threadIdx.x;
    uint64_t a1 = 0, a2 = 0;
    if (f < g.F) {
        factor_sig(g, symmetric, rv_color, f_color, seed, f, a1, a2);
        h1[f] = a1; h2[f] = a2;
    }
    block_insert(t, filter, a1, a2, f < g.F);
}

__global__ void __launch_bounds__(BLOCK) rv_insert_kernel(lhvi_graph_t g, const int32_t* __restrict__ f_color,
                                                         const int32_t* __restrict__ rv_color, uint64_t seed,
                                                         uint64_t* __restrict__ h1, uint64_t* __restrict__ h2, Table t) {
    __shared__ uint64_t filter[FILTER_SLOTS];
    for (int i = threadIdx.x; i < FILTER_SLOTS; i += BLOCK) filter[i] = EMPTY_KEY;
    __syncthreads();
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    uint64_t a = 0, b = 0;
    bool mine = false;
    if (v < g.V) {
        const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
        if (hi - lo <= HUB_DEGREE) {
            mine = true;
            a = mix64((uint64_t)(uint32_t)rv_color[v] ^ seed ^ SEED2);
            b = mix64(((uint64_t)(uint32_t)rv_color[v] + seed) * 0xA24BAED4963EE407ull + SEED1);
            for (int k = lo; k < hi; ++k) sig_terms(g, f_color, seed, k, a, b);
            a = key_of(a);
            h1[v] = a; h2[v] = b;
        }
    }
    block_insert(t, filter, a, b, mine);
}

// hubs (template variables with thousands of incident factors): a wavefront per hub.  (A workgroup per hub, also with four
// gathers in flight per thread, measured slower: 64 / 77 us against 48 us on the 10 M-edge cfg-5 graph -- the rows of the
// revenue hubs stride through the factor-major edge arrays, and more requests in flight only thrash the sectors.)
__global__ void __launch_bounds__(BLOCK) rv_insert_hub_kernel(lhvi_graph_t g, const int32_t* __restrict__ f_color,
                                                             const int32_t* __restrict__ rv_color, uint64_t seed,
                                                             uint64_t* __restrict__ h1, uint64_t* __restrict__ h2, Table t) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= (g.hub_vars ? g.n_hubs : g.V)) return;
    const int v = g.hub_vars ? g.hub_vars[i] : i;
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo <= HUB_DEGREE) return;
    uint64_t a = 0, b = 0;
    for (int k = lo + lane; k < hi; k += 64) sig_terms(g, f_color, seed, k, a, b);
    for (int off = 32; off > 0; off >>= 1) {
        a += ((uint64_t)(uint32_t)__shfl_xor((int)(a >> 32), off) << 32) | (uint32_t)__shfl_xor((int)a, off);
        b += ((uint64_t)(uint32_t)__shfl_xor((int)(b >> 32), off) << 32) | (uint32_t)__shfl_xor((int)b, off);
    }
    if (lane == 0) {
        a = key_of(a + mix64((uint64_t)(uint32_t)rv_color[v] ^ seed ^ SEED2));
        b += mix64(((uint64_t)(uint32_t)rv_color[v] + seed) * 0xA24BAED4963EE407ull + SEED1);
        h1[v] = a; h2[v] = b;
        insert_key(t, a, b);
    }
}

// exclusive scan of the bucket counts by one workgroup (BUCKETS / 1024 = 8 counts per thread)
__global__ void __launch_bounds__(1024) bucket_scan_kernel(Table t) {
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x;
    constexpr int PER = BUCKETS / 1024;
    const uint32_t* cnt = bucket_count(t);
    uint32_t* start = bucket_start(t);
    uint32_t sum = 0;
    for (int k = 0; k < PER; ++k) sum += cnt[tid * PER + k];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (int k = 0; k < PER; ++k) { start[tid * PER + k] = run; run += cnt[tid * PER + k]; }
    if (tid == 1023) start[BUCKETS] = run;
}

__global__ void __launch_bounds__(BLOCK) bucket_scatter_kernel(Table t) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= min(t.count[0], MAX_DISTINCT)) return;
    const uint64_t key = t.dkeys[i];
    const uint32_t b = (uint32_t)(key >> (64 - BUCKET_BITS));
    const uint32_t p = bucket_start(t)[b] + atomicAdd(&bucket_cursor(t)[b], 1u);
    t.skeys[p] = key; t.sslot[p] = t.dslot[i];
}

// colour of a key = keys of lower buckets + smaller keys of its own bucket = its position in numeric order
__global__ void __launch_bounds__(BLOCK) bucket_rank_kernel(Table t) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= min(t.count[0], MAX_DISTINCT)) return;
    const uint64_t key = t.skeys[p];
    const uint32_t b = (uint32_t)(key >> (64 - BUCKET_BITS));
    const uint32_t lo = bucket_start(t)[b], hi = bucket_start(t)[b + 1];
    uint32_t smaller = 0;
    for (uint32_t q = lo; q < hi; ++q) smaller += t.skeys[q] < key;
    t.rank[t.sslot[p]] = (int32_t)(lo + smaller);
}

// result: [0] number of colours, [1] fingerprint collision, [2] table overflow
__global__ void __launch_bounds__(BLOCK) assign_kernel(int n, Table t, const uint64_t* __restrict__ h1, const uint64_t* __restrict__ h2,
                                                      int32_t* __restrict__ color_out, int32_t* __restrict__ result) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) { result[0] = (int32_t)min(t.count[0], MAX_DISTINCT); if (t.count[1]) result[2] = 1; }
    if (i >= n || t.count[1]) return;
    const uint64_t key = h1[i];
    uint32_t s = home_slot(key);
    for (int probe = 0; probe < MAX_PROBES; ++probe) {
        const uint64_t cur = t.keys[s];
        if (cur == key) break;
        if (cur == EMPTY_KEY) { result[2] = 1; return; }    // cannot happen: every key was inserted
        s = (s + 1) & (TABLE_SLOTS - 1);
    }
    color_out[i] = t.rank[s];
    if (t.h2[s] != h2[i]) result[1] = 1;               // equal h1, different h2: the host retries with another seed
}

__global__ void __launch_bounds__(BLOCK) flag_kernel(int n, const uint64_t* __restrict__ key_sorted,
                                                    const uint32_t* __restrict__ idx_sorted,
                                                    const uint64_t* __restrict__ h2, int32_t* __restrict__ flag,
                                                    int32_t* __restrict__ result) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    int fl = 1;
    if (i > 0 && key_sorted[i] == key_sorted[i - 1]) {
        fl = 0;
        if (h2[idx_sorted[i]] != h2[idx_sorted[i - 1]]) result[1] = 1;   // h1 collision: host retries
    }
    flag[i] = fl;
}

__global__ void __launch_bounds__(BLOCK) scatter_rank_kernel(int n, const uint32_t* __restrict__ idx_sorted,
                                                            const int32_t* __restrict__ rank,
                                                            int32_t* __restrict__ color_out,
                                                            int32_t* __restrict__ result) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    color_out[idx_sorted[i]] = rank[i] - 1;
    if (i == n - 1) result[0] = rank[i];
}

struct Workspace {
    uint64_t *h1, *h1_sorted, *h2;
    uint32_t *idx, *idx_sorted;
    int32_t *flag, *rank;
    void* temp;
    size_t temp_bytes;
    Table table;
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t temp_bytes_for(size_t n) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                              (uint32_t*)nullptr, n, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, n, rocprim::plus<int32_t>(), (hipStream_t)0);
    return align_up(a > b ? a : b);
}

// first (smallest-index) member of every colour.  One atomicMin per item serialises on the few large clusters of a lifted
// relational graph (torch's scatter_reduce 'amin': 2.5 ms for 2.5 M items in 4 k colours; the first half million threads of a
// launch all find the table empty).  Lanes of a wavefront hold consecutive items, so among the lanes that share a colour the
// lowest one holds the smallest index: the wavefront peels its distinct colours one by one (readfirstlane of the remaining
// lanes' colour), that lane alone looks at the table -- a plain load first, the atomic only when it would lower the entry --
// and everyone else retires.  A wavefront inside one large cluster issues at most one atomic.
__global__ void __launch_bounds__(BLOCK) first_member_kernel(int n, const int32_t* __restrict__ color, int n_colors,
                                                             int32_t* __restrict__ first) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const int c = i < n ? color[i] : -1;
    if ((unsigned)c >= (unsigned)n_colors) return;
    while (true) {                                                   // (lanes that break leave the EXEC mask)
        const int c0 = __builtin_amdgcn_readfirstlane(c);            // colour of the lowest remaining lane ...
        const int i0 = __builtin_amdgcn_readfirstlane(i);            // ... which holds the smallest index of that colour here
        if (c == c0) {
            if (i == i0 && first[c] > i) atomicMin(first + c, i);
            break;
        }
    }
}

__global__ void __launch_bounds__(BLOCK) fill_i32_kernel(int n, int32_t value, int32_t* __restrict__ out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = value;
}

static size_t carve(Workspace* w, void* base, size_t n) {
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    w->h1 = (uint64_t*)take(n * 8); w->h1_sorted = (uint64_t*)take(n * 8); w->h2 = (uint64_t*)take(n * 8);
    w->idx = (uint32_t*)take(n * 4); w->idx_sorted = (uint32_t*)take(n * 4);
    w->flag = (int32_t*)take(n * 4); w->rank = (int32_t*)take(n * 4);
    w->temp_bytes = temp_bytes_for(n);
    w->temp = take(w->temp_bytes);
    Table& t = w->table;
    t.keys = (uint64_t*)take((size_t)TABLE_SLOTS * 8); t.h2 = (uint64_t*)take((size_t)TABLE_SLOTS * 8);
    t.rank = (int32_t*)take((size_t)TABLE_SLOTS * 4);
    t.dkeys = (uint64_t*)take((size_t)MAX_DISTINCT * 8); t.skeys = (uint64_t*)take((size_t)MAX_DISTINCT * 8);
    t.dslot = (uint32_t*)take((size_t)MAX_DISTINCT * 4); t.sslot = (uint32_t*)take((size_t)MAX_DISTINCT * 4);
    t.bucket = (uint32_t*)take((size_t)(3 * BUCKETS + 1) * 4);
    t.count = (uint32_t*)take(256);
    return off;
}

static int table_reset(const Table& t, hipStream_t st) {
    if (hipMemsetAsync(t.keys, 0xff, (size_t)TABLE_SLOTS * 8, st) != hipSuccess) return LHVI_E_LAUNCH;
    if (hipMemsetAsync(t.bucket, 0, (size_t)(3 * BUCKETS + 1) * 4, st) != hipSuccess) return LHVI_E_LAUNCH;
    if (hipMemsetAsync(t.count, 0, 2 * sizeof(uint32_t), st) != hipSuccess) return LHVI_E_LAUNCH;
    return LHVI_OK;
}

// distinct keys -> ranks -> colours
static int table_rank_and_assign(Workspace& w, int n, int32_t* color_out, int32_t* result, hipStream_t st) {
    Table& t = w.table;
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1024), 0, st, t);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(grid_for(MAX_DISTINCT)), dim3(BLOCK), 0, st, t);
    hipLaunchKernelGGL(bucket_rank_kernel, dim3(grid_for(MAX_DISTINCT)), dim3(BLOCK), 0, st, t);
    hipLaunchKernelGGL(assign_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, st, n, t, w.h1, w.h2, color_out, result);
    return check_launch();
}

static int rank_and_scatter(Workspace& w, int n, int32_t* color_out, int32_t* result, hipStream_t st) {
    size_t tb = w.temp_bytes;
    if (rocprim::radix_sort_pairs(w.temp, tb, w.h1, w.h1_sorted, w.idx, w.idx_sorted, (size_t)n, 0, 64, st) != hipSuccess)
        return LHVI_E_LAUNCH;
    hipLaunchKernelGGL(flag_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, st, n, w.h1_sorted, w.idx_sorted, w.h2, w.flag, result);
    tb = w.temp_bytes;
    if (rocprim::inclusive_scan(w.temp, tb, w.flag, w.rank, (size_t)n, rocprim::plus<int32_t>(), st) != hipSuccess)
        return LHVI_E_LAUNCH;
    hipLaunchKernelGGL(scatter_rank_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, st, n, w.idx_sorted, w.rank, color_out, result);
    return check_launch();
}

// sums of consecutive runs of `values` in index order: the evidence value of a cluster is the RUNNING sum of its members' values
// over its size (SuperRV.get_value, CompressedGraphWithObs.py:24-28; members in ground order here), and a library segmented
// reduction adds in a tree -- the last bit of a cluster's value then depends on who summed it.  A thread per run; a run of 256 or
// more entries (the coarse stage of a coarse-to-fine schedule holds all 250 k observed atoms of a 10 M-edge model in a handful of
// clusters) is streamed through LDS by the whole wavefront, 256 coalesced values at a time, and added up by one lane in order --
// the same additions, ~10 cycles each instead of a dependent global load each.
constexpr int SEG_LONG = 256;
__global__ void __launch_bounds__(BLOCK) segment_sum_kernel(int n_segments, const double* __restrict__ values,
                                                           const int64_t* __restrict__ offsets, double* __restrict__ out) {
    __shared__ double stage[BLOCK / WAVE][SEG_LONG];
    const int s = blockIdx.x * BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double* buf = stage[threadIdx.x >> 6];
    const int64_t lo = s < n_segments ? offsets[s] : 0, hi = s < n_segments ? offsets[s + 1] : 0;
    const bool is_long = hi - lo >= SEG_LONG;
    double acc = 0.0;
    if (!is_long) for (int64_t i = lo; i < hi; ++i) acc += values[i];
    uint64_t todo = __ballot(is_long);
    while (todo) {
        const int l = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const int64_t a = __shfl(lo, l), b = __shfl(hi, l);
        double run = 0.0;
        // the next batch's loads are in flight while lane 0 adds the current one (a batch is ~2 500 cycles of dependent additions,
        // a load round trip about as long)
        double nxt[SEG_LONG / WAVE];
#pragma unroll
        for (int k = 0; k < SEG_LONG / WAVE; ++k) { const int64_t i = a + k * WAVE + lane; nxt[k] = i < b ? values[i] : 0.0; }
        for (int64_t base = a; base < b; base += SEG_LONG) {
            const int cnt = (int)(b - base < SEG_LONG ? b - base : SEG_LONG);
#pragma unroll
            for (int k = 0; k < SEG_LONG / WAVE; ++k) buf[k * WAVE + lane] = nxt[k];
#pragma unroll
            for (int k = 0; k < SEG_LONG / WAVE; ++k) { const int64_t i = base + SEG_LONG + k * WAVE + lane; nxt[k] = i < b ? values[i] : 0.0; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                int j = 0;
                for (; j + 16 <= cnt; j += 16) {                    // sixteen LDS reads in flight, then their additions in order
                    double t[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) t[k] = buf[j + k];
#pragma unroll
                    for (int k = 0; k < 16; ++k) run += t[k];
                }
                for (; j < cnt; ++j) run += buf[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        run = __shfl(run, 0);
        if (lane == l) acc = run;
    }
    if (s < n_segments) out[s] = acc;
}

}  // namespace lhvi

using namespace lhvi;

extern "C" {

size_t lhvi_color_workspace_bytes(const lhvi_graph_t* g) {
    if (!g) return 0;
    Workspace w;
    size_t n = (size_t)(g->V > g->F ? g->V : g->F);
    return carve(&w, nullptr, n < 1 ? 1 : n);
}

// result: device int32[4] = {number of colours, collision flag (retry with another seed if 1), table overflow (repeat the
// half round with method 1), 0}
int lhvi_color_refine_factors(const lhvi_graph_t* g, const uint8_t* symmetric, const int32_t* rv_color,
                              const int32_t* f_color, int32_t* f_color_out, int32_t* n_colors_out,
                              void* ws, size_t ws_bytes, int32_t method, void* stream) {
    if (!g || !rv_color || !f_color || !f_color_out || !n_colors_out || !ws || method < 0 || method > 1) return LHVI_E_ARG;
    if (ws_bytes < lhvi_color_workspace_bytes(g)) return LHVI_E_ARG;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(n_colors_out, 0, 4 * sizeof(int32_t), st) != hipSuccess) return LHVI_E_LAUNCH;
    if (g->F == 0) return LHVI_OK;
    Workspace w;
    size_t n = (size_t)(g->V > g->F ? g->V : g->F);
    carve(&w, ws, n);
    const uint64_t seed = 0;
    if (method == 0) {
        if (int rc = table_reset(w.table, st)) return rc;
        hipLaunchKernelGGL(factor_insert_kernel, dim3(grid_for(g->F)), dim3(BLOCK), 0, st, *g, symmetric, rv_color, f_color, seed,
                           w.h1, w.h2, w.table);
        if (int rc = check_launch()) return rc;
        return table_rank_and_assign(w, g->F, f_color_out, n_colors_out, st);
    }
    hipLaunchKernelGGL(factor_sig_kernel, dim3(grid_for(g->F)), dim3(BLOCK), 0, st, *g, symmetric, rv_color, f_color,
                       seed, w.h1, w.h2, w.idx);
    if (int rc = check_launch()) return rc;
    return rank_and_scatter(w, g->F, f_color_out, n_colors_out, st);
}

int lhvi_color_refine_rvs(const lhvi_graph_t* g, const int32_t* f_color, const int32_t* rv_color,
                          int32_t* rv_color_out, int32_t* n_colors_out, void* ws, size_t ws_bytes, int32_t method, void* stream) {
    if (!g || !rv_color || !f_color || !rv_color_out || !n_colors_out || !ws || method < 0 || method > 1) return LHVI_E_ARG;
    if (ws_bytes < lhvi_color_workspace_bytes(g)) return LHVI_E_ARG;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(n_colors_out, 0, 4 * sizeof(int32_t), st) != hipSuccess) return LHVI_E_LAUNCH;
    if (g->V == 0) return LHVI_OK;
    Workspace w;
    size_t n = (size_t)(g->V > g->F ? g->V : g->F);
    carve(&w, ws, n);
    const uint64_t seed = 0;
    if (method == 0) {
        if (int rc = table_reset(w.table, st)) return rc;
        hipLaunchKernelGGL(rv_insert_kernel, dim3(grid_for(g->V)), dim3(BLOCK), 0, st, *g, f_color, rv_color, seed, w.h1, w.h2, w.table);
        const int64_t nh = g->hub_vars ? g->n_hubs : g->V;
        if (nh > 0 && g->nnz > 0)
            hipLaunchKernelGGL(rv_insert_hub_kernel, dim3(grid_for(nh * 64)), dim3(BLOCK), 0, st, *g, f_color, rv_color, seed, w.h1, w.h2, w.table);
        if (int rc = check_launch()) return rc;
        return table_rank_and_assign(w, g->V, rv_color_out, n_colors_out, st);
    }
    hipLaunchKernelGGL(rv_sig_init_kernel, dim3(grid_for(g->V)), dim3(BLOCK), 0, st, g->V, rv_color, seed, w.h1, w.h2, w.idx);
    if (g->nnz > 0)
    {
        hipLaunchKernelGGL(rv_sig_accum_kernel, dim3(grid_for(g->V)), dim3(BLOCK), 0, st, *g, f_color, seed, w.h1, w.h2);
        const int64_t nh = g->hub_vars ? g->n_hubs : g->V;
        if (nh > 0)
            hipLaunchKernelGGL(rv_sig_accum_hub_kernel, dim3(grid_for(nh * 64)), dim3(BLOCK), 0, st, *g, f_color, seed, w.h1, w.h2);
    }
    if (int rc = check_launch()) return rc;
    return rank_and_scatter(w, g->V, rv_color_out, n_colors_out, st);
}

int lhvi_color_segment_sums(const double* values, const int64_t* offsets, int32_t n_segments, double* sums_out, void* stream) {
    if (n_segments < 0 || (n_segments > 0 && (!offsets || !sums_out))) return LHVI_E_ARG;
    if (n_segments == 0) return LHVI_OK;
    hipLaunchKernelGGL(segment_sum_kernel, dim3(grid_for(n_segments)), dim3(BLOCK), 0, as_stream(stream), n_segments, values, offsets, sums_out);
    return check_launch();
}

int lhvi_color_first_members(const int32_t* color, int32_t n, int32_t n_colors, int32_t* first_out, void* stream) {
    if (n < 0 || n_colors < 0 || (n > 0 && !color) || (n_colors > 0 && !first_out)) return LHVI_E_ARG;
    hipStream_t st = as_stream(stream);
    if (n_colors == 0) return LHVI_OK;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(n_colors)), dim3(BLOCK), 0, st, n_colors, n, first_out);
    if (n > 0) hipLaunchKernelGGL(first_member_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, st, n, color, n_colors, first_out);
    return check_launch();
}

}  // extern "C"
