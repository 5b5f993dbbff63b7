// mi355sat.hip — libmi355sat.so: the C ABI of include/mi355sat.h over the HIP
// kernels in device/kernels.hip.h.  Host duties only: clause intake and
// normalisation, building the immutable clause database and the per-worker slab
// template, replicating it into HBM, launching slices of the search kernel, and
// reading verdicts / models / counters back.  There is no CPU solving path: if
// HIP is unavailable every entry point fails.
//
// Reference call sites this file serves (see include/mi355sat.h for the mapping):
//   crates/repl/src/solver_runner.rs:12-16, crates/repl/src/main.rs:295,316,329,363,
//   crates/gui/src/solver_backend.rs:78-90.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "../../include/mi355sat.h"
#include "device/kernels.hip.h"

namespace {

std::string g_new_error;
std::mutex g_new_error_mu;

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct HipErr { std::string msg; };
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess)                                                                       \
            throw HipErr{std::string(#x) + ": " + hipGetErrorString(e_)};                            \
    } while (0)

// ---- auxiliary kernels ---------------------------------------------------------
// Per-worker customisation after the template slab has been replicated:
// assumptions, scripted decisions and a worker-specific initial decision order
// (affine permutation of the variables; worker group 0 keeps the canonical order).
__global__ void ms_customize_kernel(MsLayout L, char* slabs, uint32_t n_workers, const int32_t* assump_data,
                                    const uint64_t* assump_off, const int32_t* script_data,
                                    const uint64_t* script_off, uint32_t n_instances, uint64_t seed, int32_t park_from,
                                    uint32_t wid0, int32_t phase_mix) {
    const uint32_t wid = blockIdx.x + wid0;   // workers [wid0, n_workers)
    if (wid >= n_workers) return;
    char* slab = slabs + (size_t)wid * L.slab_bytes;
    MsState* st = (MsState*)(slab + L.state);
    const uint32_t inst = n_instances ? wid % n_instances : 0;
    const uint32_t replica = n_instances ? wid / n_instances : wid;
    if (assump_off) {
        uint64_t a0 = assump_off[inst], a1 = assump_off[inst + 1];
        int32_t* dst = (int32_t*)(slab + L.assumps);
        for (uint64_t i = threadIdx.x; i < a1 - a0; i += blockDim.x) dst[i] = assump_data[a0 + i];
        if (threadIdx.x == 0) st->n_assumps = (int32_t)(a1 - a0);
    }
    if (script_off) {
        uint64_t a0 = script_off[inst], a1 = script_off[inst + 1];
        int32_t* dst = (int32_t*)(slab + L.script);
        for (uint64_t i = threadIdx.x; i < a1 - a0; i += blockDim.x) dst[i] = script_data[a0 + i];
        if (threadIdx.x == 0) st->n_script = (int32_t)(a1 - a0);
    }
    if (threadIdx.x == 0) {
        st->rng = seed * 0x9E3779B97F4A7C15ull + wid;
        if (park_from >= 0 && (int32_t)wid >= park_from) st->status = MS_ST_PARKED;  // idle until it steals a cube
    }
    if (replica > 0 && L.n_vars > 2) {
        // splitmix64 -> multiplier coprime to n_vars, offset
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(replica + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const uint32_t n = L.n_vars;
        uint32_t a = (uint32_t)(z % n) | 1u, b = (uint32_t)((z >> 32) % n);
        for (;;) {
            uint32_t x = a, y = n;
            while (y) { uint32_t t = x % y; x = y; y = t; }
            if (x == 1) break;
            a += 2;
            if (a >= n) a = 1;
        }
        int32_t* order = (int32_t*)(slab + L.vm_order);
        MsVarRec* vrec = (MsVarRec*)(slab + L.vrec);
        // initial saved phases (phase_mix): replica % 4 == 1 decides every variable TRUE first, == 2 at random;
        // the others keep FALSE (the template)
        const uint32_t pm = phase_mix > 0 ? replica & 3u : 0u;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            uint32_t v = (uint32_t)(((uint64_t)a * i + b) % n);
            order[i] = (int32_t)v;
            ((int32_t*)(slab + L.vm_pos))[v] = (int32_t)i;
            if (pm == 1) vrec[v].phase = 0;
            else if (pm == 2) vrec[v].phase = (uint8_t)((((uint64_t)v * 0x9E3779B97F4A7C15ull + z) >> 40) & 1);
        }
    }
}

// Replicates the template slab (its head and the initially used part of the watch pool) into
// every worker slab with 16-byte copies: one launch instead of two memcpys per worker.
__global__ void ms_replicate_kernel(const char* tmpl, char* slabs, uint64_t slab_bytes, uint64_t head_bytes,
                                    uint64_t pool_off, uint64_t pool_bytes) {
    char* dst = slabs + (size_t)blockIdx.y * slab_bytes;
    const uint64_t n_head = head_bytes / 16, n_pool = (pool_bytes + 15) / 16;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_head + n_pool; i += stride) {
        const uint64_t off = i < n_head ? i * 16 : pool_off + (i - n_head) * 16;
        *(uint4*)(dst + off) = *(const uint4*)(tmpl + off);
    }
}

// Applies the scheduler's decisions between two slices: update u = (worker, status, restart_req,
// n_assumps, data offset); the worker's assumption list (its cube) is rewritten from data[].
__global__ void ms_assign_kernel(MsLayout L, char* slabs, uint32_t n_upd, const int32_t* upd, const int32_t* data) {
    const uint32_t u = blockIdx.x;
    if (u >= n_upd) return;
    const int32_t worker = upd[5 * u], status = upd[5 * u + 1], restart = upd[5 * u + 2], n = upd[5 * u + 3],
                  off = upd[5 * u + 4];
    char* slab = slabs + (size_t)worker * L.slab_bytes;
    MsState* st = (MsState*)(slab + L.state);
    int32_t* dst = (int32_t*)(slab + L.assumps);
    for (int32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = data[off + i];
    if (threadIdx.x == 0) {
        st->n_assumps = n;
        st->status = status;
        if (restart) st->restart_req = 1;
        st->n_split = 0;
    }
}

__global__ void ms_gather_states_kernel(MsLayout L, const char* slabs, uint32_t n_workers, MsState* out) {
    const uint32_t wid = blockIdx.x * blockDim.x + threadIdx.x;
    if (wid >= n_workers) return;
    out[wid] = *(const MsState*)(slabs + (size_t)wid * L.slab_bytes + L.state);
}

// Between two slices: move the clauses the workers exported during the last slice into the global
// ring, once each (a 64-bit order-independent signature in a hash set filters what many workers
// learnt alike).  One thread per worker; nothing else runs on the stream meanwhile.
__device__ inline unsigned long long share_mix(unsigned long long x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__global__ void ms_share_collect_kernel(MsLayout L, char* slabs, uint32_t n_workers, int32_t* pool, uint32_t slots,
                                        unsigned long long* share_n, unsigned long long* hash, uint32_t hash_mask,
                                        uint32_t* intake, uint32_t intake_cap, int ordered) {
    // ordered (opts.deterministic): ONE thread takes the workers in index order, so ring order, the hash set's races and the
    // intake cut-off do not depend on timing
    uint32_t wid = blockIdx.x * blockDim.x + threadIdx.x;
    if (ordered) {
        if (wid != 0) return;
    } else if (wid >= n_workers) return;
  for (; wid < n_workers; wid++) {
    char* slab = slabs + (size_t)wid * L.slab_bytes;
    MsState* st = (MsState*)(slab + L.state);
    const uint32_t n = st->exp_n;
    if (!n) { if (ordered) continue; return; }
    const int32_t* exp = (const int32_t*)(slab + L.exp);
    for (uint32_t r = 0; r < n && r < MS_EXPORT_RECS; r++) {
        const int32_t* rec = exp + r * MS_SHARE_REC;
        const int sz = rec[0] & 63;
        if (sz < 1 || sz > MS_SHARE_MAXLEN) continue;
        unsigned long long h = (unsigned long long)sz * 0xd6e8feb86659fd93ull;
        for (int j = 0; j < sz; j++) h += share_mix((unsigned long long)(uint32_t)rec[1 + j]);
        if (!h) h = 1;
        bool fresh = true;
        uint32_t i = (uint32_t)(h >> 20) & hash_mask;
        for (int probe = 0; probe < 16; probe++) {
            unsigned long long prev = atomicCAS(&hash[i], 0ull, h);
            if (prev == 0ull) break;
            if (prev == h) { fresh = false; break; }
            i = (i + 1) & hash_mask;
        }
        if (!fresh) continue;
        // every worker attaches every record: units and binaries always pass, of the longer clauses only
        // intake_cap per collection (first come), so that attaching stays a small part of a slice
        if (sz > 2 && atomicAdd(intake, 1u) >= intake_cap) continue;
        const unsigned long long slot = atomicAdd(share_n, 1ull) % slots;
        int32_t* dst = pool + slot * MS_SHARE_REC;
        for (int j = 0; j <= sz; j++) dst[j] = rec[j];
    }
    st->exp_n = 0;
    if (!ordered) return;
  }
}

// Subsumption and self-subsuming resolution (the other half of what `simp::Glucose` does before search; its
// variable elimination is not restated: measured to remove < 15 % of these encodings' variables).  One thread per
// clause C (literals sorted): every clause D in the occurrence list of C's rarest literal p is merged against C -
// C inside D: D is subsumed;  C inside D except ONE literal x that D has negated: D loses ~x (the resolvent of C and
// D on x subsumes D).  The list of ~p finds the D's that lose ~p.  sig = 64-bit signature over VARIABLES
// (sig(C) & ~sig(D) != 0 rules D out without touching its literals).
__device__ inline int subsume_check(const int32_t* lits, const uint64_t* offs, uint32_t c, uint32_t d) {
    // 0 no, 1 C subsumes D, 2 + x: D can drop literal ~x  (returned as 2 + (~x))
    uint64_t i = offs[c], ie = offs[c + 1], j = offs[d], je = offs[d + 1];
    int flip = -1;
    for (; i < ie; i++) {
        const int32_t a = lits[i];
        while (j < je && (lits[j] >> 1) < (a >> 1)) j++;
        if (j == je || (lits[j] >> 1) != (a >> 1)) return 0;
        if (lits[j] != a) {
            if (flip >= 0) return 0;
            flip = lits[j];
        }
        j++;
    }
    return flip < 0 ? 1 : 2 + flip;
}
__global__ void ms_subsume_kernel(uint32_t n_clauses, const int32_t* lits, const uint64_t* offs, const unsigned long long* sig,
                                  const uint32_t* occ_off, const uint32_t* occ, uint32_t max_size, uint32_t* subsumed,
                                  int32_t* drop_lit) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clauses) return;
    const uint32_t sz = (uint32_t)(offs[c + 1] - offs[c]);
    if (sz > max_size) return;
    int32_t p = lits[offs[c]];
    uint32_t best = 0xffffffffu;
    for (uint64_t k = offs[c]; k < offs[c + 1]; k++) {
        const int32_t l = lits[k];
        const uint32_t n = (occ_off[l + 1] - occ_off[l]) + (occ_off[(l ^ 1) + 1] - occ_off[l ^ 1]);
        if (n < best) { best = n; p = l; }
    }
    const unsigned long long sc = sig[c];
    for (int side = 0; side < 2; side++) {
        const int32_t q = side ? (p ^ 1) : p;
        for (uint32_t e = occ_off[q]; e < occ_off[q + 1]; e++) {
            const uint32_t d = occ[e];
            if (d == c || (sc & ~sig[d]) != 0) continue;
            const uint32_t dz = (uint32_t)(offs[d + 1] - offs[d]);
            if (dz < sz) continue;
            const int r = subsume_check(lits, offs, c, d);
            if (r == 1) { if (dz > sz || c < d) subsumed[d] = 1; }          // equal clauses: the lower index stays
            else if (r >= 2) atomicMax((int*)&drop_lit[d], r - 2);          // one literal per clause and pass (the largest: the choice does not depend on timing)
        }
    }
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        release();
        n = count;
        if (count) HIPCHK(hipMalloc((void**)&p, count * sizeof(T)));
    }
    void upload(const std::vector<T>& v, hipStream_t s) {
        alloc(v.size());
        if (!v.empty()) HIPCHK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// The worker slabs are by far the largest allocation (4096 x 37 MiB = 147 GiB at rect 64x64) and hipMalloc
// of that size takes 2-5 s on MI355X (measured; 33 GB/s) - more than most solves of the refinement loop, which
// makes a FRESH solver per bound (main.rs:295).  So a freed handle parks its slab buffer here, one per device,
// and the next handle of the process takes it over if it is large enough.  mi355sat_release_cached_memory()
// returns it to the driver.
struct SlabCache {
    std::mutex mu;
    char* p[64] = {nullptr};
    size_t bytes[64] = {0};
};
SlabCache g_slab_cache;

struct SlabBuf {
    char* p = nullptr;
    size_t n = 0;        // bytes in use
    size_t cap = 0;      // bytes allocated
    int dev = 0;
    static size_t cached_bytes(int dev) {
        std::lock_guard<std::mutex> g(g_slab_cache.mu);
        return dev >= 0 && dev < 64 ? g_slab_cache.bytes[dev] : 0;
    }
    void alloc(size_t bytes, int device) {
        release();
        dev = device;
        if (!bytes) return;
        {
            std::lock_guard<std::mutex> g(g_slab_cache.mu);
            if (dev >= 0 && dev < 64 && g_slab_cache.p[dev]) {
                if (g_slab_cache.bytes[dev] >= bytes) { p = g_slab_cache.p[dev]; cap = g_slab_cache.bytes[dev]; }
                else (void)hipFree(g_slab_cache.p[dev]);     // too small: make room before the larger request
                g_slab_cache.p[dev] = nullptr;
                g_slab_cache.bytes[dev] = 0;
            }
        }
        if (!p) { HIPCHK(hipMalloc((void**)&p, bytes)); cap = bytes; }
        n = bytes;
    }
    void release() {
        if (p) {
            std::lock_guard<std::mutex> g(g_slab_cache.mu);
            if (dev >= 0 && dev < 64 && g_slab_cache.bytes[dev] < cap) {
                if (g_slab_cache.p[dev]) (void)hipFree(g_slab_cache.p[dev]);
                g_slab_cache.p[dev] = p;
                g_slab_cache.bytes[dev] = cap;
            } else (void)hipFree(p);
        }
        p = nullptr;
        n = cap = 0;
    }
    ~SlabBuf() { release(); }
};

}  // namespace

// One eliminated variable: literal x of it and the clauses (elim_lits[begin, end), -1 terminated, x included) that held x.
struct MsElim { int32_t x; uint32_t begin, end; };

struct mi355sat {
    mi355sat_opts opts{};
    int device = 0;
    // clause intake (DIMACS literals)
    std::vector<int32_t> lits;
    std::vector<uint64_t> offs{0};
    std::vector<int32_t> pending;
    uint64_t max_var = 0;
    // results
    std::vector<int8_t> model;                 // plain solve
    std::vector<std::vector<int8_t>> batch_models;
    mi355sat_stats_t stats{};
    std::string err;
    std::string proof_path;
    // interrupt flag: pinned host memory the kernels poll
    int32_t* stop_flag = nullptr;
    std::atomic<int> interrupted{0};
    // device
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevBuf<int32_t> d_cl_lits, d_bin_lits, d_tern_owner;
    DevBuf<ms_int2> d_tern_pairs;
    bool lds_val = false;                      // assignment staged in LDS (2 bits/var)
    uint32_t lds_val_bytes = 0;
    DevBuf<char> d_template;
    SlabBuf d_slabs;
    DevBuf<MsState> d_states;
    DevBuf<int32_t> d_any_done, d_assump, d_script, d_proof;
    DevBuf<uint32_t> d_proof_len;
    uint32_t proof_cap = 0;                    // words per worker in d_proof
    FILE* proof_file = nullptr;                // open while a solve logs its DRUP proof
    DevBuf<uint64_t> d_assump_off, d_script_off;
    // learnt-clause exchange between workers (layout.h MS_SHARE_*)
    DevBuf<int32_t> d_share_pool;
    DevBuf<unsigned long long> d_share_n, d_share_hash;
    DevBuf<uint32_t> d_share_intake;
    uint32_t share_slots = 0, share_hash_n = 0;
    uint64_t share_slices = 0;
    uint64_t share_export_pos = 0;             // ring records below this were handed out by mi355sat_share_export already
    MsShared sh{};
    MsLayout L{};
    uint32_t n_workers = 0;
    uint32_t n_alloc = 0;                      // workers [0, n_alloc) have their slab (the rest is allocated on demand)
    uint64_t pool_init = 0;                    // watch-pool entries in use in the template
    // prepared formula facts
    bool trivially_unsat = false;
    std::vector<int8_t> fixed;                 // per var: 0 free, 1 true, -1 false (level-0 facts)
    uint32_t n_vars = 0;
    std::vector<uint32_t> perm;                // caller's variable index -> device variable index (Prepared::perm)
    std::vector<int32_t> subst;                // per caller variable: the literal (2*var + neg) that replaced it, or itself
    std::vector<int32_t> simp_proof;           // DRUP lemmas of the simplification (internal literals, -1 terminated)
    std::vector<MsElim> elims;                 // variables eliminated before search and the clauses that rebuild their values
    std::vector<int32_t> elim_lits;
    struct SweepHolder* sweep = nullptr;        // stepwise sweep in progress (mi355sat_sweep_*)
};

namespace {

// ---- formula preparation ---------------------------------------------------------
struct Prepared {
    uint32_t n_vars = 0;
    bool unsat = false;
    std::vector<int32_t> units;                 // internal literals fixed at level 0 (trail prefix)
    std::vector<uint32_t> perm;                 // caller's variable index (0-based) -> device variable index
    bool units_propagated = false;              // true if simplification ran (queue starts empty)
    std::vector<MsClauseHdr> cl_hdr;            // long clauses (>= 4 literals)
    std::vector<int32_t> cl_lits;               // each clause 16-byte aligned, padded with its first literal
    std::vector<MsLitHdr> lit_hdr;              // 2*n_vars
    std::vector<int32_t> bin_lits;
    std::vector<ms_int2> tern_pairs;
    std::vector<int32_t> tern_owner;
};

inline int32_t to_internal(int32_t d) { return d > 0 ? 2 * (d - 1) : 2 * (-d - 1) + 1; }
inline int32_t to_device(const std::vector<uint32_t>& perm, int32_t d) {   // DIMACS literal -> device literal
    return d > 0 ? 2 * (int32_t)perm[d - 1] : 2 * (int32_t)perm[-d - 1] + 1;
}

// Variable order for the device: every per-variable array (the 2-bit assignment: 256 variables per
// 64-byte line; records, watch and list headers: 4 per line) is gathered by variable index, so
// variables that meet in clauses should be neighbours.  The caller's numbering is by encoder family;
// this one grows blobs of 256 variables breadth-first through the clauses (variable - clause -
// variable), each new blob seeded on the border of an earlier one, so a blob is a compact patch of the
// variable interaction graph (for the support grids: a few neighbouring tiles with all their
// variables, or a subtree of the totalizer).  Fixed and unused variables go last.
void locality_order(uint32_t nv, const std::vector<int32_t>& nl, const std::vector<uint64_t>& no,
                    const std::vector<int8_t>& val, std::vector<uint32_t>& perm) {
    const uint32_t NONE = 0xffffffffu, BLOB = 256;
    const size_t nn = no.size() - 1;
    perm.assign(nv, NONE);
    std::vector<uint32_t> occ_off((size_t)nv + 1, 0);
    for (size_t c = 0; c < nn; c++) {
        if (no[c + 1] - no[c] > 64) continue;
        for (uint64_t k = no[c]; k < no[c + 1]; k++) occ_off[(nl[k] >> 1) + 1]++;
    }
    for (size_t i = 0; i < nv; i++) occ_off[i + 1] += occ_off[i];
    std::vector<uint32_t> occ(occ_off[nv]), fill(occ_off.begin(), occ_off.end() - 1);
    for (size_t c = 0; c < nn; c++) {
        if (no[c + 1] - no[c] > 64) continue;
        for (uint64_t k = no[c]; k < no[c + 1]; k++) occ[fill[nl[k] >> 1]++] = (uint32_t)c;
    }
    std::vector<uint32_t> vstamp(nv, 0), cstamp(nn, 0), frontier, q;
    std::vector<uint8_t> pending(nv, 0);
    size_t fhead = 0;
    uint32_t next = 0, blob = 0, scan = 0;
    for (;;) {
        uint32_t seed = NONE;
        while (fhead < frontier.size()) { uint32_t v = frontier[fhead++]; if (perm[v] == NONE) { seed = v; break; } }
        if (seed == NONE) {
            while (scan < nv && (perm[scan] != NONE || val[scan] != 0 || occ_off[scan] == occ_off[scan + 1])) scan++;
            if (scan == nv) break;
            seed = scan;
        }
        blob++;
        q.clear();
        q.push_back(seed);
        vstamp[seed] = blob;
        size_t h = 0;
        uint32_t cnt = 0;
        while (h < q.size() && cnt < BLOB) {
            const uint32_t v = q[h++];
            perm[v] = next++;
            cnt++;
            for (uint32_t e = occ_off[v]; e < occ_off[v + 1]; e++) {
                const uint32_t c = occ[e];
                if (cstamp[c] == blob) continue;
                cstamp[c] = blob;
                for (uint64_t k = no[c]; k < no[c + 1]; k++) {
                    const uint32_t u = (uint32_t)(nl[k] >> 1);
                    if (perm[u] != NONE || vstamp[u] == blob || val[u] != 0) continue;
                    vstamp[u] = blob;
                    q.push_back(u);
                }
            }
        }
        for (; h < q.size(); h++)   // the blob is full: its border seeds later blobs
            if (!pending[q[h]]) { pending[q[h]] = 1; frontier.push_back(q[h]); }
    }
    for (uint32_t v = 0; v < nv; v++) if (perm[v] == NONE) perm[v] = next++;
}

// The caller's clauses in normal form (internal literals 2*var + neg, caller's numbering; sorted, no duplicate
// literals, no tautologies, no units): what the simplification steps work on.
struct Formula {
    uint32_t nv = 0;
    bool unsat = false;
    std::vector<int32_t> nl;                 // literals
    std::vector<uint64_t> no{0};             // clause offsets
    std::vector<int8_t> val;                 // level-0 facts: 0 unassigned, 1 true, -1 false
    std::vector<int32_t> units;              // the same as a list, in derivation order
    size_t units_done = 0;                   // units[0..units_done) have been propagated through nl / no
    std::vector<int32_t> subst;              // per variable: the literal that replaces it (2*v = itself)
    std::vector<int32_t> proof;              // DRUP lemmas justifying units / equivalences / rewritten clauses (-1 terminated)
    bool log_proof = false;
    std::vector<int32_t> frozen_lits;        // internal literals (caller's numbering) whose variables must survive: the assumptions
    std::vector<MsElim> elims;               // eliminated variables in elimination order + the clauses that rebuild their value
    std::vector<int32_t> elim_lits;
    uint64_t n_failed = 0, n_necessary = 0, n_equiv = 0, n_subsumed = 0, n_strengthened = 0, n_eliminated = 0;
    size_t n_clauses() const { return no.size() - 1; }
    int8_t lv(int32_t l) const { int8_t v = val[l >> 1]; return (l & 1) ? (int8_t)-v : v; }
    bool assign_unit(int32_t l) {            // false on contradiction
        const int8_t want = (l & 1) ? -1 : 1;
        int8_t& v = val[l >> 1];
        if (v == 0) { v = want; units.push_back(l); return true; }
        return v == want;
    }
    void lemma(std::initializer_list<int32_t> c) { if (log_proof) { proof.insert(proof.end(), c); proof.push_back(-1); } }
    void lemma(const std::vector<int32_t>& c) { if (log_proof) { proof.insert(proof.end(), c.begin(), c.end()); proof.push_back(-1); } }
};

// Sort / dedupe / drop tautologies of one clause; units go to F.val.  Returns false if the clause is dropped.
bool normal_clause(Formula& F, std::vector<int32_t>& tmp) {
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    for (size_t i = 0; i + 1 < tmp.size(); i++) if ((tmp[i] ^ 1) == tmp[i + 1]) return false;
    if (tmp.empty()) { F.unsat = true; return false; }
    if (tmp.size() == 1) { if (!F.assign_unit(tmp[0])) F.unsat = true; return false; }
    return true;
}

void normalise(const mi355sat& s, Formula& F) {
    F.nv = (uint32_t)s.max_var;
    F.val.assign(F.nv, 0);
    F.subst.resize(F.nv);
    for (uint32_t v = 0; v < F.nv; v++) F.subst[v] = 2 * (int32_t)v;
    const size_t nc = s.offs.size() - 1;
    F.nl.reserve(s.lits.size());
    std::vector<int32_t> tmp;
    for (size_t c = 0; c < nc && !F.unsat; c++) {
        tmp.clear();
        for (uint64_t k = s.offs[c]; k < s.offs[c + 1]; k++) tmp.push_back(to_internal(s.lits[k]));
        if (!normal_clause(F, tmp)) continue;
        F.nl.insert(F.nl.end(), tmp.begin(), tmp.end());
        F.no.push_back(F.nl.size());
    }
}

// Level-0 unit propagation over occurrence lists, then strip satisfied clauses and false literals.
void propagate_units(Formula& F) {
    if (F.unsat || F.units_done == F.units.size()) return;
    const uint32_t nv = F.nv;
    const size_t nn = F.n_clauses();
    std::vector<uint32_t> occ_off(2 * (size_t)nv + 1, 0);
    for (int32_t l : F.nl) occ_off[l + 1]++;
    for (size_t i = 0; i < 2 * (size_t)nv; i++) occ_off[i + 1] += occ_off[i];
    std::vector<uint32_t> occ(F.nl.size()), fill(occ_off.begin(), occ_off.end() - 1);
    for (size_t c = 0; c < nn; c++)
        for (uint64_t k = F.no[c]; k < F.no[c + 1]; k++) occ[fill[F.nl[k]]++] = (uint32_t)c;
    size_t qh = F.units_done;
    while (qh < F.units.size() && !F.unsat) {
        const int32_t f = F.units[qh++] ^ 1;
        for (uint32_t e = occ_off[f]; e < occ_off[f + 1] && !F.unsat; e++) {
            const uint32_t c = occ[e];
            int32_t unit = -1;
            int nfree = 0;
            bool sat = false;
            for (uint64_t k = F.no[c]; k < F.no[c + 1]; k++) {
                const int8_t v = F.lv(F.nl[k]);
                if (v > 0) { sat = true; break; }
                if (v == 0) { nfree++; unit = F.nl[k]; }
            }
            if (sat) continue;
            if (nfree == 0) F.unsat = true;
            else if (nfree == 1 && !F.assign_unit(unit)) F.unsat = true;
        }
    }
    if (F.unsat) return;
    F.units_done = F.units.size();
    std::vector<int32_t> nl2;
    std::vector<uint64_t> no2{0};
    nl2.reserve(F.nl.size());
    for (size_t c = 0; c < nn; c++) {
        bool sat = false;
        const size_t start = nl2.size();
        for (uint64_t k = F.no[c]; k < F.no[c + 1]; k++) {
            const int8_t v = F.lv(F.nl[k]);
            if (v > 0) { sat = true; break; }
            if (v == 0) nl2.push_back(F.nl[k]);
        }
        if (sat) { nl2.resize(start); continue; }
        no2.push_back(nl2.size());
    }
    F.nl.swap(nl2);
    F.no.swap(no2);
}

// Device form of the formula: variable order, binary / ternary CSRs, long clauses.
void build_csr(const mi355sat& s, const Formula& F, bool units_propagated, Prepared& P) {
    const uint32_t nv = F.nv;
    P.n_vars = nv;
    P.unsat = F.unsat;
    P.units = F.units;
    P.units_propagated = units_propagated;
    if (P.unsat) return;
    std::vector<int32_t> nl = F.nl;
    const std::vector<uint64_t>& no = F.no;
    const size_t nn = F.n_clauses();
    // device variable order
    if (s.opts.var_order > 0) locality_order(nv, nl, no, F.val, P.perm);
    else { P.perm.resize(nv); for (uint32_t v = 0; v < nv; v++) P.perm[v] = v; }
    for (auto& l : nl) l = 2 * (int32_t)P.perm[l >> 1] | (l & 1);
    for (auto& l : P.units) l = 2 * (int32_t)P.perm[l >> 1] | (l & 1);
    // split: binary -> implication CSR, ternary -> pair CSR, >= 4 literals -> watched clauses
    std::vector<std::pair<int32_t, int32_t>> bins;
    std::vector<std::array<int32_t, 3>> terns;
    for (size_t c = 0; c < nn; c++) {
        uint64_t len = no[c + 1] - no[c];
        if (len == 2) bins.push_back({nl[no[c]], nl[no[c] + 1]});
        else if (len == 3) terns.push_back({nl[no[c]], nl[no[c] + 1], nl[no[c] + 2]});
        else {
            P.cl_hdr.push_back(MsClauseHdr{(uint32_t)P.cl_lits.size(), (uint32_t)len});
            P.cl_lits.insert(P.cl_lits.end(), nl.begin() + no[c], nl.begin() + no[c + 1]);
            while (P.cl_lits.size() % 4) P.cl_lits.push_back(nl[no[c]]);
        }
    }
    for (int k = 0; k < 4; k++) P.cl_lits.push_back(0);  // a lane may read one 16-byte group past the last clause
    for (auto& b : bins) if (b.first > b.second) std::swap(b.first, b.second);
    std::sort(bins.begin(), bins.end());
    bins.erase(std::unique(bins.begin(), bins.end()), bins.end());
    for (auto& c : terns) std::sort(c.begin(), c.end());
    std::sort(terns.begin(), terns.end());
    terns.erase(std::unique(terns.begin(), terns.end()), terns.end());
    P.lit_hdr.assign(2 * (size_t)nv, MsLitHdr{0, 0, 0, 0});
    for (auto& b : bins) { P.lit_hdr[b.first ^ 1].bin_n++; P.lit_hdr[b.second ^ 1].bin_n++; }
    for (auto& c : terns)
        for (int k = 0; k < 3; k++) P.lit_hdr[c[k] ^ 1].tern_n++;
    {
        uint32_t bo = 0, to = 0;
        for (auto& hd : P.lit_hdr) { hd.bin_off = bo; bo += hd.bin_n; hd.tern_off = to; to += hd.tern_n; }
    }
    P.bin_lits.resize(2 * bins.size());
    {
        std::vector<uint32_t> fill(2 * (size_t)nv);
        for (size_t i = 0; i < fill.size(); i++) fill[i] = P.lit_hdr[i].bin_off;
        for (auto& b : bins) {
            P.bin_lits[fill[b.first ^ 1]++] = b.second;   // ~a -> b
            P.bin_lits[fill[b.second ^ 1]++] = b.first;   // ~b -> a
        }
    }
    P.tern_pairs.resize(3 * terns.size());
    P.tern_owner.resize(3 * terns.size());
    {
        std::vector<uint32_t> fill(2 * (size_t)nv);
        for (size_t i = 0; i < fill.size(); i++) fill[i] = P.lit_hdr[i].tern_off;
        for (auto& c : terns)
            for (int k = 0; k < 3; k++) {
                uint32_t e = fill[c[k] ^ 1]++;          // list of the literal that makes c[k] false
                P.tern_pairs[e] = ms_int2{c[(k + 1) % 3], c[(k + 2) % 3]};
                P.tern_owner[e] = c[k] ^ 1;
            }
    }
}

void prepare(const mi355sat& s, bool simplify, Prepared& P) {
    Formula F;
    normalise(s, F);
    if (simplify) propagate_units(F);
    build_csr(s, F, simplify, P);
}

// ---- slab template -----------------------------------------------------------------
void build_layout_and_template(mi355sat& s, const Prepared& P, uint32_t assump_cap, uint32_t script_cap,
                               std::vector<char>& tmpl) {
    const uint32_t nv = P.n_vars, no = (uint32_t)P.cl_hdr.size();
    MsLayout& L = s.L;
    memset(&L, 0, sizeof L);
    L.n_vars = nv;
    L.n_orig = no;
    // capacities
    const uint64_t base_lits = P.cl_lits.size() + 3 * P.tern_pairs.size() / 3;
    L.learnt_cap = (uint32_t)std::min<uint64_t>(1u << 17, std::max<uint64_t>(1u << 15, 2 * (uint64_t)no + 4096));
    // (round 2: 2 M literals at most instead of 4 M - a store that runs full reduces early, add_learnt / on_conflict)
    L.learnt_lit_cap = (uint32_t)std::min<uint64_t>(2u << 20, std::max<uint64_t>(1u << 19, 6 * base_lits));
    L.vm_cap = 3 * nv + 256;
    L.assump_cap = assump_cap;
    L.script_cap = script_cap;
    // watch lists: literal t's list holds the clauses currently watching ~t.  Initial slots get
    // 50% + 2 entries of slack; a list that outgrows its slot moves to the top of the bump pool and
    // the device compacts the pool again at every learnt-clause reduction (rebuild_watches).
    std::vector<uint32_t> cap(2 * (size_t)nv, 0);
    for (uint32_t c = 0; c < no; c++) { cap[P.cl_lits[P.cl_hdr[c].start] ^ 1]++; cap[P.cl_lits[P.cl_hdr[c].start + 1] ^ 1]++; }
    uint64_t pool_need = 0;
    std::vector<uint32_t> base(2 * (size_t)nv);
    for (size_t t = 0; t < cap.size(); t++) {
        cap[t] += (cap[t] >> 1) + 2;
        base[t] = (uint32_t)pool_need;
        pool_need += cap[t];
    }
    // room for a dense rebuild with every learnt slot in use, plus 50% for relocations in between
    uint64_t dense_max = 3 * ((uint64_t)no + L.learnt_cap) + 4 * (uint64_t)nv;   // (rebuild_watches: size + size/2 + 2 per list)
    uint64_t pool_cap = dense_max + dense_max / 2 + (1u << 16);
    if (pool_cap > 0xfffffff0ull) throw HipErr{"formula too large (watch pool)"};
    L.pool_cap = (uint32_t)pool_cap;
    size_t off = 0;
    auto place = [&](uint64_t& field, size_t bytes) { field = off; off = align_up(off + bytes, 256); };
    place(L.state, sizeof(MsState));
    place(L.val, 16 * (((size_t)nv + 15) / 16));
    place(L.vrec, sizeof(MsVarRec) * (size_t)nv);
    place(L.vm_pos, 4 * (size_t)nv);
    place(L.best, (size_t)nv);
    place(L.trail, 4 * (size_t)nv);
    place(L.trail_lim, 4 * ((size_t)nv + 1));
    place(L.vm_order, 4 * (size_t)L.vm_cap);
    place(L.wl, sizeof(MsClauseRec) * ((size_t)no + L.learnt_cap));
    place(L.whdr, sizeof(MsWatchHdr) * 2 * (size_t)nv);
    place(L.lc_lbd, 4 * (size_t)L.learnt_cap);
    place(L.learnt_buf, 4 * ((size_t)nv + 1));
    place(L.toclear, 4 * ((size_t)nv + 1));
    place(L.lvl_stamp, 4 * ((size_t)nv + 2));
    place(L.remap, 4 * (size_t)L.learnt_cap);
    place(L.overflow, 4 * 3 * MS_OVERFLOW_CAP);
    place(L.assumps, 4 * (size_t)std::max<uint32_t>(assump_cap, 1));
    place(L.script, 4 * (size_t)std::max<uint32_t>(script_cap, 1));
    place(L.exp, 4 * (size_t)MS_EXPORT_RECS * MS_SHARE_REC);
    // the big, cold-tailed arrays last
    place(L.lc_lits, 4 * (size_t)L.learnt_lit_cap);
    place(L.pool, 16 * (size_t)L.pool_cap);
    L.slab_bytes = align_up(off, 4096);

    // Only the head of the slab (everything before lc_lits) plus the initial watch
    // pool needs initial contents; lc_lits is left as allocated.
    tmpl.assign(L.slab_bytes, 0);
    char* T = tmpl.data();
    MsState* st = (MsState*)(T + L.state);
    st->status = MS_ST_RUNNING;
    st->trail_n = (int32_t)P.units.size();
    st->qhead = P.units_propagated ? st->trail_n : 0;
    st->n_levels = 0;
    st->vm_end = (int32_t)nv;
    st->vm_search = (int32_t)nv - 1;
    st->pool_top = (uint32_t)pool_need;
    s.pool_init = pool_need;
    st->next_reduce = s.opts.reduce_first > 0 ? (uint64_t)s.opts.reduce_first : 2000;
    st->next_rephase = 2000;
    st->next_vivify = 1500;   // (easy bounds are decided before that: vivification is for the long refutations)
    memset(T + L.best, 255, nv);
    uint8_t* val = (uint8_t*)(T + L.val);        // zero = every variable unassigned
    MsVarRec* vrec = (MsVarRec*)(T + L.vrec);
    int32_t* vm_order = (int32_t*)(T + L.vm_order);
    // initial decision order: the caller's numbering (the search starts at position nv - 1); variables that occur in no
    // clause (replaced by an equivalent literal, eliminated) behind all others
    std::vector<uint8_t> occurs(nv, 0);
    for (const MsClauseHdr& h : P.cl_hdr)      // (cl_lits carries padding behind the last clause)
        for (uint32_t k = 0; k < h.size; k++) occurs[P.cl_lits[h.start + k] >> 1] = 1;
    for (size_t t = 0; t < P.lit_hdr.size(); t++) if (P.lit_hdr[t].bin_n || P.lit_hdr[t].tern_n) occurs[t >> 1] = 1;
    uint32_t pos = nv;
    for (int pass = 0; pass < 2; pass++)
        for (uint32_t e = 0; e < nv; e++) {
            const uint32_t v = P.perm[e];
            if ((occurs[v] != 0) != (pass == 0)) continue;
            pos--;
            vrec[v] = MsVarRec{0, MS_REASON_NONE, 0, 0, /*phase=*/1, /*seen=*/0};
            ((int32_t*)(T + L.vm_pos))[v] = (int32_t)pos;
            vm_order[pos] = (int32_t)v;
        }
    int32_t* trail = (int32_t*)(T + L.trail);
    for (size_t i = 0; i < P.units.size(); i++) {
        int32_t l = P.units[i];
        trail[i] = l;
        val[l >> 1] = (uint8_t)(2u | (uint32_t)(l & 1));
    }
    MsClauseRec* wl = (MsClauseRec*)(T + L.wl);
    MsWatchHdr* whdr = (MsWatchHdr*)(T + L.whdr);
    int4* pool = (int4*)(T + L.pool);
    for (size_t t = 0; t < cap.size(); t++)
        whdr[MS_HIDX(t, nv)] = MsWatchHdr{base[t], 0, cap[t], 0, P.lit_hdr[t].bin_off, P.lit_hdr[t].bin_n, P.lit_hdr[t].tern_off, P.lit_hdr[t].tern_n};
    for (uint32_t c = 0; c < no; c++) {
        int32_t a = P.cl_lits[P.cl_hdr[c].start], b = P.cl_lits[P.cl_hdr[c].start + 1];
        wl[c] = MsClauseRec{a, b, P.cl_hdr[c].start, P.cl_hdr[c].size};
        pool[whdr[MS_HIDX(a ^ 1, nv)].base + whdr[MS_HIDX(a ^ 1, nv)].size++] = make_int4((int)c, b, (int)P.cl_hdr[c].start, (int)P.cl_hdr[c].size);
        pool[whdr[MS_HIDX(b ^ 1, nv)].base + whdr[MS_HIDX(b ^ 1, nv)].size++] = make_int4((int)c, a, (int)P.cl_hdr[c].start, (int)P.cl_hdr[c].size);
    }
}

void set_error(mi355sat* s, const std::string& m) { s->err = m; }

void upload_formula(mi355sat& s, const Prepared& P, uint32_t assump_cap, uint32_t script_cap, uint32_t want_workers,
                    uint32_t initial_workers = 0) {
    HIPCHK(hipSetDevice(s.device));
    std::vector<char> tmpl;
    build_layout_and_template(s, P, assump_cap, script_cap, tmpl);
    s.n_vars = P.n_vars;
    s.perm = P.perm;
    s.d_cl_lits.upload(P.cl_lits, s.stream);
    s.d_bin_lits.upload(P.bin_lits.empty() ? std::vector<int32_t>{0} : P.bin_lits, s.stream);
    s.d_tern_pairs.upload(P.tern_pairs.empty() ? std::vector<ms_int2>{ms_int2{0, 0}} : P.tern_pairs, s.stream);
    s.d_tern_owner.upload(P.tern_owner.empty() ? std::vector<int32_t>{0} : P.tern_owner, s.stream);
    s.sh.n_vars = P.n_vars;
    s.sh.n_orig = (uint32_t)P.cl_hdr.size();
    s.sh.cl_lits = s.d_cl_lits.p;
    s.sh.bin_lits = s.d_bin_lits.p;
    s.sh.tern_pairs = s.d_tern_pairs.p;
    s.sh.tern_owner = s.d_tern_owner.p;
    // assignment in LDS (2 bits per variable) when it still leaves room for 12 waves per CU
    // (measured on rect 64x64: 12 waves/CU with the assignment in HBM beat 6 waves/CU with it in LDS)
    s.lds_val_bytes = ((P.n_vars + 15) / 16) * 4 + 5 * ((P.n_vars + 31) / 32) * 4;   // 2-bit assignment + five 1-bit maps (marks, current level, level 0, minimisation: failed, queued)
    s.lds_val = s.opts.lds_val == 1 || (s.opts.lds_val == 0 && s.lds_val_bytes <= 10 * 1024);
    if (s.lds_val_bytes > 150 * 1024) s.lds_val = false;
    // worker count limited by free HBM
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    free_b += SlabBuf::cached_bytes(s.device);   // a parked slab buffer of an earlier handle is ours to reuse
    uint64_t fit = (uint64_t)((double)free_b * 0.85) / (s.L.slab_bytes * 1ull);
    if (fit < 2) throw HipErr{"not enough device memory for one worker slab"};
    uint32_t W = (uint32_t)std::min<uint64_t>(want_workers, fit - 1);
    if (W == 0) W = 1;
    s.n_workers = W;
    s.d_template.alloc(s.L.slab_bytes);
    HIPCHK(hipMemcpyAsync(s.d_template.p, tmpl.data(), s.L.slab_bytes, hipMemcpyHostToDevice, s.stream));
    // Slabs for `initial_workers` only (the ramp-up's first phase) unless a parked buffer already covers all of
    // them: hipMalloc costs ~30 ms per GiB, and an easy instance never needs the rest (grow_workers).
    uint32_t A = W;
    if (initial_workers && initial_workers < W && SlabBuf::cached_bytes(s.device) < (size_t)W * s.L.slab_bytes) A = initial_workers;
    s.n_alloc = A;
    const double t_alloc0 = now_s();
    s.d_slabs.alloc((size_t)A * s.L.slab_bytes, s.device);
    if (s.opts.verbose) fprintf(stderr, "[mi355sat] slab allocation %.1f GiB (%u of %u workers): %.3f s\n", (double)A * s.L.slab_bytes / 1073741824.0, A, W, now_s() - t_alloc0);
    s.d_states.alloc(W);
    s.d_any_done.alloc(1);
    // clause exchange: on unless switched off, whenever there is more than one worker
    s.share_slots = 0;
    if (s.opts.share >= 0 && W > 1 && script_cap == 0) {
        s.share_slots = 1u << 19;   // 64 MiB of records
        while ((uint64_t)s.share_slots < (uint64_t)W * MS_EXPORT_RECS) s.share_slots <<= 1;   // one collection (<= W * MS_EXPORT_RECS
                                                                                              // records) never wraps onto itself
        s.share_hash_n = 1u << 22;
        s.d_share_pool.alloc((size_t)s.share_slots * MS_SHARE_REC);
        s.d_share_n.alloc(1);
        s.d_share_hash.alloc(s.share_hash_n);
        s.d_share_intake.alloc(1);
    }
    HIPCHK(hipStreamSynchronize(s.stream));
    if (s.opts.verbose)
        fprintf(stderr, "[mi355sat] vars=%u long=%u tern=%zu bin=%zu units=%zu slab=%.2f MiB workers=%u lds_val=%d\n", P.n_vars,
                s.sh.n_orig, P.tern_pairs.size() / 3, P.bin_lits.size() / 2, P.units.size(), s.L.slab_bytes / 1048576.0, W, (int)s.lds_val);
}

// Replicate the template into every worker slab (head + initial watch pool only).
void replicate_template(mi355sat& s, uint32_t from, uint32_t to) {   // workers [from, to)
    const MsLayout& L = s.L;
    const size_t head = L.lc_lits;  // everything before the learnt literal store
    if (to <= from) return;
    hipLaunchKernelGGL(ms_replicate_kernel, dim3(64, to - from), dim3(256), 0, s.stream, (const char*)s.d_template.p,
                       s.d_slabs.p + (size_t)from * L.slab_bytes, (uint64_t)L.slab_bytes, (uint64_t)head, (uint64_t)L.pool,
                       (uint64_t)(16 * s.pool_init));
    HIPCHK(hipGetLastError());
}

void reset_workers(mi355sat& s) {
    replicate_template(s, 0, s.n_alloc);
    HIPCHK(hipMemsetAsync(s.d_any_done.p, 0, sizeof(int32_t), s.stream));
    if (s.share_slots) {
        HIPCHK(hipMemsetAsync(s.d_share_n.p, 0, sizeof(unsigned long long), s.stream));
        HIPCHK(hipMemsetAsync(s.d_share_hash.p, 0, sizeof(unsigned long long) * s.share_hash_n, s.stream));
        s.share_slices = 0;
        s.share_export_pos = 0;
    }
}

void customize(mi355sat& s, const std::vector<int32_t>* assump, const std::vector<uint64_t>* assump_off,
               const std::vector<int32_t>* script, const std::vector<uint64_t>* script_off, uint32_t n_instances,
               int32_t park_from = -1) {
    if (assump_off) {
        s.d_assump.upload(assump->empty() ? std::vector<int32_t>{0} : *assump, s.stream);
        s.d_assump_off.upload(*assump_off, s.stream);
    }
    if (script_off) {
        s.d_script.upload(script->empty() ? std::vector<int32_t>{0} : *script, s.stream);
        s.d_script_off.upload(*script_off, s.stream);
    }
    hipLaunchKernelGGL(ms_customize_kernel, dim3(s.n_alloc), dim3(256), 0, s.stream, s.L, s.d_slabs.p,
                       s.n_alloc, assump_off ? s.d_assump.p : nullptr, assump_off ? s.d_assump_off.p : nullptr,
                       script_off ? s.d_script.p : nullptr, script_off ? s.d_script_off.p : nullptr, n_instances,
                       s.opts.seed, park_from, 0u, s.opts.phase_mix);
    HIPCHK(hipGetLastError());
}

// The search outlived the ramp-up's first phase: give the remaining workers their slabs.  The running workers'
// slabs move into the full-size buffer (device-to-device copy), the new ones start from the template with the
// assumption list of instance w % n_instances, as at the beginning.
void grow_workers(mi355sat& s, uint32_t n_instances, uint32_t target) {
    target = std::min(target, s.n_workers);
    if (s.n_alloc >= target) return;
    const uint32_t old = s.n_alloc;
    const double t0 = now_s();
    if (s.d_slabs.cap >= (size_t)target * s.L.slab_bytes) {
        // the buffer came from the parked one of an earlier handle (or of this handle's probing) and has the room
        // already: no allocation (hipMalloc of 8.7 GiB is 0.25 s - a quarter of the rect 24x24 ladder), no move
        s.d_slabs.n = (size_t)target * s.L.slab_bytes;
    } else {
        SlabBuf big;
        big.alloc((size_t)target * s.L.slab_bytes, s.device);
        HIPCHK(hipMemcpyAsync(big.p, s.d_slabs.p, (size_t)old * s.L.slab_bytes, hipMemcpyDeviceToDevice, s.stream));
        HIPCHK(hipStreamSynchronize(s.stream));
        std::swap(big.p, s.d_slabs.p); std::swap(big.n, s.d_slabs.n); std::swap(big.cap, s.d_slabs.cap); std::swap(big.dev, s.d_slabs.dev);
        big.release();   // the small buffer (parked if nothing larger is)
    }
    s.n_alloc = target;
    replicate_template(s, old, s.n_alloc);
    hipLaunchKernelGGL(ms_customize_kernel, dim3(s.n_alloc - old), dim3(256), 0, s.stream, s.L, s.d_slabs.p, s.n_alloc,
                       s.d_assump.p, s.d_assump_off.p, (const int32_t*)nullptr, (const uint64_t*)nullptr, n_instances, s.opts.seed,
                       -1, old, s.opts.phase_mix);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s.stream));
    if (s.opts.verbose) fprintf(stderr, "[mi355sat] grew from %u to %u worker slabs (%.1f GiB): %.3f s\n", old, s.n_alloc,
                                (double)s.n_alloc * s.L.slab_bytes / 1073741824.0, now_s() - t0);
}

void gather_states(mi355sat& s, std::vector<MsState>& out) {
    out.assign(s.n_workers, MsState{});   // a worker without a slab yet: RUNNING, all counters zero
    hipLaunchKernelGGL(ms_gather_states_kernel, dim3((s.n_alloc + 63) / 64), dim3(64), 0, s.stream, s.L,
                       s.d_slabs.p, s.n_alloc, s.d_states.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out.data(), s.d_states.p, sizeof(MsState) * s.n_alloc, hipMemcpyDeviceToHost, s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
}

// ---- formula simplification before search (`simp::Glucose`, crates/repl/src/main.rs:17) --------------------------
// Equivalent-literal substitution: strongly connected components of the binary implication graph (iterative
// Tarjan on the host - a few 10^5 edges), every literal replaced by the smallest literal of its component.
uint64_t els_scc(Formula& F) {
    if (F.unsat) return 0;
    const uint32_t nl2 = 2 * F.nv;
    const size_t nn = F.n_clauses();
    std::vector<uint32_t> off(nl2 + 1, 0);
    for (size_t c = 0; c < nn; c++)
        if (F.no[c + 1] - F.no[c] == 2) { off[(F.nl[F.no[c]] ^ 1) + 1]++; off[(F.nl[F.no[c] + 1] ^ 1) + 1]++; }
    for (uint32_t i = 0; i < nl2; i++) off[i + 1] += off[i];
    std::vector<int32_t> adj(off[nl2]);
    {
        std::vector<uint32_t> fill(off.begin(), off.end() - 1);
        for (size_t c = 0; c < nn; c++)
            if (F.no[c + 1] - F.no[c] == 2) {
                const int32_t a = F.nl[F.no[c]], b = F.nl[F.no[c] + 1];
                adj[fill[a ^ 1]++] = b;
                adj[fill[b ^ 1]++] = a;
            }
    }
    std::vector<int32_t> index(nl2, -1), low(nl2, 0), rep(nl2, -1), stack, work, it(nl2, 0);
    std::vector<uint8_t> on(nl2, 0);
    int32_t counter = 0;
    for (uint32_t root = 0; root < nl2; root++) {
        if (index[root] >= 0 || off[root] == off[root + 1]) continue;
        work.push_back((int32_t)root);
        while (!work.empty()) {
            const int32_t v = work.back();
            if (index[v] < 0) { index[v] = low[v] = counter++; stack.push_back(v); on[v] = 1; it[v] = (int32_t)off[v]; }
            bool descended = false;
            while (it[v] < (int32_t)off[v + 1]) {
                const int32_t u = adj[it[v]++];
                if (index[u] < 0) { work.push_back(u); descended = true; break; }
                if (on[u]) low[v] = std::min(low[v], index[u]);
            }
            if (descended) continue;
            if (low[v] == index[v]) {
                size_t k = stack.size();
                int32_t m = v;
                do { k--; m = std::min(m, stack[k]); } while (stack[k] != v);
                for (size_t j = k; j < stack.size(); j++) { rep[stack[j]] = m; on[stack[j]] = 0; }
                stack.resize(k);
            }
            work.pop_back();
            if (!work.empty()) low[work.back()] = std::min(low[work.back()], low[v]);
        }
    }
    uint64_t n_sub = 0;
    std::vector<int32_t> map(nl2);
    for (uint32_t l = 0; l < nl2; l++) map[l] = (int32_t)l;
    for (uint32_t v = 0; v < F.nv; v++) {
        const int32_t r = rep[2 * v];
        if (r < 0 || r == (int32_t)(2 * v)) continue;
        if (r == (int32_t)(2 * v + 1)) { F.unsat = true; return 0; }      // x and ~x in one component
        if ((r >> 1) > (int32_t)v) continue;                              // (the mirror component decides)
        map[2 * v] = r;
        map[2 * v + 1] = r ^ 1;
        F.subst[v] = r;
        F.lemma({(int32_t)(2 * v + 1), r});
        F.lemma({(int32_t)(2 * v), r ^ 1});
        n_sub++;
    }
    if (!n_sub) return 0;
    F.n_equiv += n_sub;
    std::vector<int32_t> nl2v, tmp;
    std::vector<uint64_t> no2{0};
    nl2v.reserve(F.nl.size());
    for (size_t c = 0; c < nn && !F.unsat; c++) {
        tmp.clear();
        bool changed = false;
        for (uint64_t k = F.no[c]; k < F.no[c + 1]; k++) { tmp.push_back(map[F.nl[k]]); changed = changed || map[F.nl[k]] != F.nl[k]; }
        if (changed) {
            std::vector<int32_t> lem = tmp;
            std::sort(lem.begin(), lem.end());
            lem.erase(std::unique(lem.begin(), lem.end()), lem.end());
            bool taut = false;
            for (size_t i = 0; i + 1 < lem.size(); i++) taut = taut || (lem[i] ^ 1) == lem[i + 1];
            if (!taut) F.lemma(lem);
        }
        if (!normal_clause(F, tmp)) continue;
        nl2v.insert(nl2v.end(), tmp.begin(), tmp.end());
        no2.push_back(nl2v.size());
    }
    F.nl.swap(nl2v);
    F.no.swap(no2);
    return n_sub;
}

void launch_probe(mi355sat& s);   // below (needs launch parameters)
const char* status_text(int st);

// Failed-literal probing on the device (ms_probe_kernel): both polarities of every variable that still occurs.
uint64_t device_probe(mi355sat& s, Formula& F) {
    if (F.unsat || F.n_clauses() == 0) return 0;
    Prepared P;
    build_csr(s, F, /*units_propagated=*/true, P);
    std::vector<uint8_t> occurs(F.nv, 0);
    for (int32_t l : F.nl) occurs[l >> 1] = 1;
    std::vector<uint32_t> cand;
    for (uint32_t v = 0; v < F.nv; v++) if (occurs[v] && !F.val[v]) cand.push_back(v);
    if (cand.empty()) return 0;
    uint32_t W = (uint32_t)std::min<size_t>(1024, cand.size());
    std::vector<std::vector<int32_t>> per;
    std::vector<int32_t> script;
    std::vector<uint64_t> soff;
    uint32_t cap = 0;
    for (;;) {   // as many workers as the device has room for (the script length is part of the slab layout)
        per.assign(W, {});
        for (size_t i = 0; i < cand.size(); i++) {
            const int32_t dl = 2 * (int32_t)P.perm[cand[i]];
            per[i % W].push_back(dl);
            per[i % W].push_back(dl ^ 1);
        }
        script.clear();
        soff.assign(1, 0);
        cap = 0;
        for (auto& v : per) { script.insert(script.end(), v.begin(), v.end()); soff.push_back(script.size()); cap = std::max<uint32_t>(cap, (uint32_t)v.size()); }
        upload_formula(s, P, 0, cap, W);
        if (s.n_workers >= W) break;
        W = s.n_workers;
    }
    reset_workers(s);
    customize(s, nullptr, nullptr, &script, &soff, W);
    launch_probe(s);
    // results
    std::vector<uint32_t> inv(P.perm.size());
    for (uint32_t e = 0; e < P.perm.size(); e++) inv[P.perm[e]] = e;
    auto to_caller = [&](int32_t dl) { return 2 * (int32_t)inv[dl >> 1] | (dl & 1); };
    const size_t fact_cap = 3 * (((size_t)P.n_vars + 1) / 3);
    std::vector<int32_t> res((size_t)W * cap), nfacts(W);
    HIPCHK(hipMemcpy2D(res.data(), 4 * (size_t)cap, s.d_slabs.p + s.L.script, s.L.slab_bytes, 4 * (size_t)cap, W, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy2D(nfacts.data(), 4, s.d_slabs.p + s.L.learnt_buf, s.L.slab_bytes, 4, W, hipMemcpyDeviceToHost));
    // only as many columns of the fact rows as the busiest worker filled (a full row is n_vars ints per worker)
    size_t fact_used = 0;
    for (uint32_t w = 0; w < W; w++) fact_used = std::max(fact_used, 3 * (size_t)std::max(nfacts[w], 0));
    const size_t fact_row = std::min(fact_cap, fact_used);
    std::vector<int32_t> facts((size_t)W * std::max<size_t>(fact_row, 1));
    if (fact_row) HIPCHK(hipMemcpy2D(facts.data(), 4 * fact_row, s.d_slabs.p + s.L.toclear, s.L.slab_bytes, 4 * fact_row, W, hipMemcpyDeviceToHost));
    std::vector<MsState> sts;
    gather_states(s, sts);
    uint64_t n_new = 0;
    for (uint32_t w = 0; w < W && !F.unsat; w++) {
        if (sts[w].status < 0) throw HipErr{std::string("probing: ") + status_text(sts[w].status)};
        if (sts[w].status == MS_ST_UNSAT) { F.unsat = true; break; }       // conflict among the formula's own units
        for (size_t d = 0; d < per[w].size(); d++)
            if (res[(size_t)w * cap + d] == -1) {                          // failed literal: its negation is a fact
                const int32_t a = to_caller(per[w][d]);
                if (F.val[a >> 1] == 0) { F.lemma({a ^ 1}); F.n_failed++; n_new++; }
                if (!F.assign_unit(a ^ 1)) F.unsat = true;
            }
        for (int32_t f = 0; f < nfacts[w] && 3 * ((size_t)f + 1) <= fact_row && !F.unsat; f++) {
            const int32_t* t = facts.data() + (size_t)w * fact_row + 3 * (size_t)f;
            const int32_t m = to_caller(t[1]), a = to_caller(t[2]);
            if (t[0] == 1) {                                               // a -> m and ~a -> m
                if (F.val[m >> 1] == 0) { F.lemma({a ^ 1, m}); F.lemma({a, m}); F.lemma({m}); F.n_necessary++; n_new++; }
                if (!F.assign_unit(m)) F.unsat = true;
            } else if (t[0] == 2 && (m >> 1) != (a >> 1)) {                // a -> m and ~a -> ~m: m == a, as two binary clauses
                std::vector<int32_t> c1{a ^ 1, m}, c2{a, m ^ 1};
                for (auto* c : {&c1, &c2}) {
                    F.lemma(*c);
                    if (normal_clause(F, *c)) { F.nl.insert(F.nl.end(), c->begin(), c->end()); F.no.push_back(F.nl.size()); }
                }
                n_new++;
            }
        }
    }
    return n_new;
}

// Subsumption / self-subsuming resolution on the device (ms_subsume_kernel), applied on the host.
uint64_t device_subsume(mi355sat& s, Formula& F) {
    if (F.unsat || F.n_clauses() == 0) return 0;
    const size_t nn = F.n_clauses();
    std::vector<unsigned long long> sig(nn, 0);
    std::vector<uint32_t> occ_off(2 * (size_t)F.nv + 1, 0);
    for (size_t c = 0; c < nn; c++)
        for (uint64_t k = F.no[c]; k < F.no[c + 1]; k++) {
            sig[c] |= 1ull << (((uint32_t)(F.nl[k] >> 1) * 2654435761u) >> 26);
            occ_off[F.nl[k] + 1]++;
        }
    for (size_t i = 0; i < 2 * (size_t)F.nv; i++) occ_off[i + 1] += occ_off[i];
    std::vector<uint32_t> occ(F.nl.size()), fill(occ_off.begin(), occ_off.end() - 1);
    for (size_t c = 0; c < nn; c++)
        for (uint64_t k = F.no[c]; k < F.no[c + 1]; k++) occ[fill[F.nl[k]]++] = (uint32_t)c;
    DevBuf<int32_t> d_lits, d_drop;
    DevBuf<uint64_t> d_offs;
    DevBuf<unsigned long long> d_sig;
    DevBuf<uint32_t> d_occ_off, d_occ, d_sub;
    d_lits.upload(F.nl, s.stream);
    d_offs.upload(F.no, s.stream);
    d_sig.upload(sig, s.stream);
    d_occ_off.upload(occ_off, s.stream);
    d_occ.upload(occ, s.stream);
    d_sub.alloc(nn);
    d_drop.alloc(nn);
    HIPCHK(hipMemsetAsync(d_sub.p, 0, 4 * nn, s.stream));
    HIPCHK(hipMemsetAsync(d_drop.p, 0xff, 4 * nn, s.stream));
    hipLaunchKernelGGL(ms_subsume_kernel, dim3((uint32_t)((nn + 255) / 256)), dim3(256), 0, s.stream, (uint32_t)nn, d_lits.p, d_offs.p,
                       d_sig.p, d_occ_off.p, d_occ.p, 64u, d_sub.p, d_drop.p);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> sub(nn);
    std::vector<int32_t> drop(nn);
    HIPCHK(hipMemcpyAsync(sub.data(), d_sub.p, 4 * nn, hipMemcpyDeviceToHost, s.stream));
    HIPCHK(hipMemcpyAsync(drop.data(), d_drop.p, 4 * nn, hipMemcpyDeviceToHost, s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    uint64_t n = 0;
    std::vector<int32_t> nl2, tmp;
    std::vector<uint64_t> no2{0};
    nl2.reserve(F.nl.size());
    for (size_t c = 0; c < nn && !F.unsat; c++) {
        if (sub[c]) { F.n_subsumed++; n++; continue; }
        tmp.assign(F.nl.begin() + F.no[c], F.nl.begin() + F.no[c + 1]);
        if (drop[c] >= 0) {
            tmp.erase(std::remove(tmp.begin(), tmp.end(), drop[c]), tmp.end());
            F.lemma(tmp);
            F.n_strengthened++;
            n++;
            if (!normal_clause(F, tmp)) continue;
        }
        nl2.insert(nl2.end(), tmp.begin(), tmp.end());
        no2.push_back(nl2.size());
    }
    F.nl.swap(nl2);
    F.no.swap(no2);
    return n;
}

inline bool bve_enabled(const mi355sat& s) { return s.opts.simp == 2; }

// Bounded variable elimination - `SimpSolver::eliminate` of the reference's backend ([ext] Een & Biere 2005, MiniSat's
// limits: no more clauses than before (grow = 0), no resolvent longer than 20 literals).  A variable x that is not
// frozen (assumptions are) is resolved away: every clause with x against every clause with ~x, tautologies dropped;
// if that does not make the formula larger, the resolvents replace the clauses of x.  The clauses of the smaller side
// are kept aside (F.elims): in reverse elimination order they give x its value in a model of the rest
// (`extendModel`).  Host code: occurrence lists and a cost-ordered queue, a few 10 ms on these formulas; every
// resolvent is a RUP lemma of the proof.
uint64_t bve_eliminate(Formula& F) {
    if (F.unsat || F.n_clauses() == 0) return 0;
    const uint32_t nv = F.nv;
    const int CLAUSE_LIM = 20, OCC_LIM = 400;
    std::vector<uint8_t> frozen(nv, 0), gone(nv, 0);
    for (int32_t l : F.frozen_lits) {
        while (F.subst[l >> 1] != 2 * (l >> 1)) l = F.subst[l >> 1] ^ (l & 1);
        frozen[l >> 1] = 1;
    }
    std::vector<int32_t>& nl = F.nl;
    std::vector<uint64_t>& no = F.no;
    std::vector<uint8_t> alive(F.n_clauses(), 1);
    std::vector<std::vector<uint32_t>> occ(2 * (size_t)nv);
    std::vector<uint32_t> n_occ(2 * (size_t)nv, 0);
    for (size_t c = 0; c < F.n_clauses(); c++)
        for (uint64_t k = no[c]; k < no[c + 1]; k++) { occ[nl[k]].push_back((uint32_t)c); n_occ[nl[k]]++; }
    auto cost = [&](uint32_t v) { return (uint64_t)n_occ[2 * v] * n_occ[2 * v + 1]; };
    typedef std::pair<uint64_t, uint32_t> QE;
    std::priority_queue<QE, std::vector<QE>, std::greater<QE>> heap;
    for (uint32_t v = 0; v < nv; v++)
        if (!frozen[v] && !F.val[v] && (n_occ[2 * v] || n_occ[2 * v + 1])) heap.push({cost(v), v});
    std::vector<uint32_t> stamp(2 * (size_t)nv, 0);
    uint32_t epoch = 0;
    std::vector<uint32_t> P, N;
    std::vector<int32_t> res;
    uint64_t n_elim = 0;
    auto kill = [&](uint32_t c) {
        alive[c] = 0;
        for (uint64_t k = no[c]; k < no[c + 1]; k++) {
            const int32_t l = nl[k];
            n_occ[l]--;
            const uint32_t u = (uint32_t)(l >> 1);
            if (!frozen[u] && !gone[u] && !F.val[u]) heap.push({cost(u), u});
        }
    };
    while (!heap.empty() && !F.unsat) {
        const QE top = heap.top();
        heap.pop();
        const uint32_t v = top.second;
        if (gone[v] || F.val[v] || top.first != cost(v)) continue;     // stale entry (a fresher one is in the queue)
        const int32_t x = 2 * (int32_t)v, nx = x + 1;
        if (n_occ[x] + n_occ[nx] == 0) continue;
        if (n_occ[x] > (uint32_t)OCC_LIM || n_occ[nx] > (uint32_t)OCC_LIM || top.first > 4096) continue;
        P.clear();
        N.clear();
        for (uint32_t c : occ[x]) if (alive[c]) P.push_back(c);
        for (uint32_t c : occ[nx]) if (alive[c]) N.push_back(c);
        occ[x] = P;
        occ[nx] = N;
        // count the non-tautological resolvents
        size_t cnt = 0;
        bool ok = true;
        for (size_t i = 0; i < P.size() && ok; i++) {
            epoch++;
            const uint32_t c = P[i];
            const int clen = (int)(no[c + 1] - no[c]);
            for (uint64_t k = no[c]; k < no[c + 1]; k++) stamp[nl[k]] = epoch;
            for (size_t j = 0; j < N.size() && ok; j++) {
                const uint32_t d = N[j];
                int extra = 0;
                bool taut = false;
                for (uint64_t k = no[d]; k < no[d + 1] && !taut; k++) {
                    const int32_t l = nl[k];
                    if (l == nx) continue;
                    if (stamp[l ^ 1] == epoch) taut = true;
                    else if (stamp[l] != epoch) extra++;
                }
                if (taut) continue;
                if (++cnt > P.size() + N.size() || clen - 1 + extra > CLAUSE_LIM) ok = false;
            }
        }
        if (!ok) continue;
        // eliminate: keep the smaller side for the model, add the resolvents, drop both sides
        {
            const bool pos_side = P.size() <= N.size();
            const std::vector<uint32_t>& side = pos_side ? P : N;
            MsElim e{pos_side ? x : nx, (uint32_t)F.elim_lits.size(), 0};
            for (uint32_t c : side) {
                F.elim_lits.insert(F.elim_lits.end(), nl.begin() + no[c], nl.begin() + no[c + 1]);
                F.elim_lits.push_back(-1);
            }
            e.end = (uint32_t)F.elim_lits.size();
            F.elims.push_back(e);
        }
        gone[v] = 1;
        for (size_t i = 0; i < P.size() && !F.unsat; i++)
            for (size_t j = 0; j < N.size() && !F.unsat; j++) {
                const uint32_t c = P[i], d = N[j];
                res.clear();
                for (uint64_t k = no[c]; k < no[c + 1]; k++) if (nl[k] != x) res.push_back(nl[k]);
                for (uint64_t k = no[d]; k < no[d + 1]; k++) if (nl[k] != nx) res.push_back(nl[k]);
                std::sort(res.begin(), res.end());
                res.erase(std::unique(res.begin(), res.end()), res.end());
                bool taut = false;
                for (size_t q = 0; q + 1 < res.size(); q++) taut = taut || (res[q] ^ 1) == res[q + 1];
                if (taut) continue;
                F.lemma(res);
                if (res.size() <= 1) {
                    if (res.empty() || !F.assign_unit(res[0])) F.unsat = true;
                    continue;
                }
                const uint32_t id = (uint32_t)F.n_clauses();
                nl.insert(nl.end(), res.begin(), res.end());
                no.push_back(nl.size());
                alive.push_back(1);
                for (int32_t l : res) { occ[l].push_back(id); n_occ[l]++; }
            }
        for (uint32_t c : P) kill(c);
        for (uint32_t c : N) kill(c);
        n_elim++;
    }
    if (!n_elim) return 0;
    F.n_eliminated += n_elim;
    std::vector<int32_t> nl2;
    std::vector<uint64_t> no2{0};
    nl2.reserve(nl.size());
    for (size_t c = 0; c < F.n_clauses(); c++) {
        if (!alive[c]) continue;
        nl2.insert(nl2.end(), nl.begin() + no[c], nl.begin() + no[c + 1]);
        no2.push_back(nl2.size());
    }
    F.nl.swap(nl2);
    F.no.swap(no2);
    return n_elim;
}

// The values of the eliminated variables in a model of the remaining formula (MiniSat's extendModel): last eliminated
// first; x is false unless one of its kept clauses has every other literal false.  model: 1 true / -1 false per variable.
void extend_model(const std::vector<MsElim>& elims, const std::vector<int32_t>& elim_lits, std::vector<int8_t>& model) {
    for (size_t i = elims.size(); i-- > 0;) {
        const MsElim& e = elims[i];
        const size_t v = (size_t)(e.x >> 1);
        if (v >= model.size()) continue;
        bool need = false;
        for (uint32_t k = e.begin; k < e.end && !need;) {
            bool others_false = true;
            for (; elim_lits[k] >= 0; k++) {
                const int32_t l = elim_lits[k];
                if (l == e.x || (size_t)(l >> 1) >= model.size()) continue;
                const int8_t m = model[l >> 1];
                if (((l & 1) ? -m : m) > 0) others_false = false;
            }
            k++;
            need = others_false;
        }
        model[v] = (int8_t)(((e.x & 1) != 0) == need ? -1 : 1);
    }
}

// The whole pipeline: units, then rounds of {equivalent literals, probing} while they find something, then
// subsumption, then variable elimination (and subsumption among its resolvents).  Everything it derives is a
// consequence of the caller's formula alone (never of assumptions); the assumptions' variables are not eliminated.
void simplify_formula(mi355sat& s, Formula& F) {
    propagate_units(F);
    if (s.opts.simp < 0 || F.unsat) return;
    const double t0 = now_s();
    const size_t c0 = F.n_clauses(), l0 = F.nl.size(), u0 = F.units.size();
    double t_els = 0, t_probe = 0, t_sub = 0;
    for (int round = 0; round < 3 && !F.unsat; round++) {
        double ta = now_s();
        uint64_t n = els_scc(F);
        propagate_units(F);
        t_els += now_s() - ta;
        ta = now_s();
        n += device_probe(s, F);
        propagate_units(F);
        t_probe += now_s() - ta;
        if (!n) break;
    }
    for (int pass = 0; pass < 3 && !F.unsat; pass++) {
        const double ta = now_s();
        const uint64_t n = device_subsume(s, F);
        propagate_units(F);
        t_sub += now_s() - ta;
        if (!n) break;
    }
    if (s.opts.verbose)
        fprintf(stderr, "[mi355sat] simplification stages: equivalent literals %.3f s, probing %.3f s, subsumption %.3f s\n", t_els, t_probe, t_sub);
    const double t1 = now_s();
    if (bve_enabled(s) && !F.unsat && bve_eliminate(F)) {
        propagate_units(F);
        for (int pass = 0; pass < 2 && !F.unsat; pass++) {
            const uint64_t n = device_subsume(s, F);
            propagate_units(F);
            if (!n) break;
        }
    }
    if (s.opts.verbose && bve_enabled(s))
        fprintf(stderr, "[mi355sat] variable elimination %.3f s: %llu variables\n", now_s() - t1, (unsigned long long)F.n_eliminated);
    if (s.opts.verbose)
        fprintf(stderr, "[mi355sat] simplification %.3f s: clauses %zu -> %zu, literals %zu -> %zu, units +%zu (failed literals %llu, necessary %llu), "
                "equivalent variables %llu, subsumed %llu, strengthened %llu%s\n", now_s() - t0, c0, F.n_clauses(), l0, F.nl.size(),
                F.units.size() - u0, (unsigned long long)F.n_failed, (unsigned long long)F.n_necessary, (unsigned long long)F.n_equiv,
                (unsigned long long)F.n_subsumed, (unsigned long long)F.n_strengthened, F.unsat ? " - UNSAT" : "");
}

void accumulate_stats(mi355sat& s, const std::vector<MsState>& sts) {
    mi355sat_stats_t& o = s.stats;
    uint64_t learnts = 0, llits = 0;
    uint64_t props = 0, dec = 0, confl = 0, rest = 0, red = 0, nw = 0, ncl = 0, nm = 0, ne = 0, steps = 0, redo = 0;
    for (auto& st : sts) {
        steps += st.n_steps; redo += st.n_redo;
        props += st.propagations; dec += st.decisions; confl += st.conflicts; rest += st.restarts;
        red += st.reduce_dbs; nw += st.n_watch; ncl += st.n_cl_lit; nm += st.n_move; ne += st.n_enq;
        learnts += st.n_learnts; llits += st.lc_lits_n;
    }
    o.propagations += props; o.decisions += dec; o.conflicts += confl; o.restarts += rest; o.reduce_dbs += red;
    o.n_deq += props; o.n_watch += nw; o.n_cl_lit += ncl; o.n_move += nm; o.n_enq += ne;
    o.learnts = learnts; o.learnt_literals = llits;
    o.bcp_steps += steps; o.bcp_requeued += redo;
    uint64_t exported = 0, imported = 0, imported_units = 0;
    for (auto& st : sts) { exported += st.n_exported; imported += st.n_imported; imported_units += st.n_imported_units; }
    o.shared_exported += exported; o.shared_imported += imported; o.shared_imported_units += imported_units;
    uint64_t prof[16] = {0}, cyc = 0;
    for (auto& st : sts) { for (int i = 0; i < 16; i++) prof[i] += st.prof[i]; cyc += st.slice_cycles; }
    if (prof[0] && s.opts.verbose) {
        static const char* nm[] = {"offsets", "binary", "ternary", "long", "close", "analyze", "backjump+learn", "decide", "reduce"};
        fprintf(stderr, "[mi355sat] phase cycle shares of %.3e worker-cycles:", (double)cyc);
        for (int i = 0; i < 9; i++) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * (double)prof[i] / (double)cyc);
        uint64_t confl = 0, ll = 0, lt = 0;
        for (auto& st : sts) { confl += st.conflicts; ll += st.learnt_lits_total; lt += st.learnt_total; }
        fprintf(stderr, "; resolution steps per conflict %.1f, learnt clause %.1f literals", (double)prof[9] / (double)std::max<uint64_t>(1, confl),
                (double)ll / (double)std::max<uint64_t>(1, lt));
        fprintf(stderr, "; of analyze: recursive minimisation %.1f%%, local %.1f%%; %.1f nodes per call, %.2f calls per conflict\n",
                100.0 * (double)prof[10] / (double)cyc, 100.0 * (double)prof[11] / (double)cyc,
                (double)prof[12] / (double)std::max<uint64_t>(1, prof[13]), (double)prof[13] / (double)std::max<uint64_t>(1, confl));
    }
}


void fetch_model(mi355sat& s, uint32_t worker, std::vector<int8_t>& out, uint64_t n_vars_out) {
    std::vector<uint8_t> asg((size_t)s.n_vars + 1);
    if (s.n_vars)
        HIPCHK(hipMemcpy(asg.data(), s.d_slabs.p + (size_t)worker * s.L.slab_bytes + s.L.val, s.n_vars, hipMemcpyDeviceToHost));
    // the device's values, then the eliminated variables (their kept clauses mention device variables and variables
    // eliminated later only), then the variables replaced by an equivalent literal (representatives have smaller indices)
    std::vector<int8_t> m(s.n_vars, 0);
    for (uint64_t v = 0; v < s.n_vars; v++) m[v] = asg[s.perm[v]] == MS_ASG_TRUE ? 1 : -1;  // (a variable left free would read false)
    extend_model(s.elims, s.elim_lits, m);
    for (uint64_t v = 0; v < s.n_vars; v++) {
        const int32_t r = v < s.subst.size() ? s.subst[v] : 2 * (int32_t)v;
        if (r != 2 * (int32_t)v && (uint64_t)(r >> 1) < v) m[v] = (r & 1) ? (int8_t)-m[r >> 1] : m[r >> 1];
    }
    out.assign(n_vars_out, 0);
    for (uint64_t v = 0; v < n_vars_out && v < s.n_vars; v++) out[v] = m[v];
}

struct SliceResult { float ms; };

SliceResult launch_slice(mi355sat& s, int mode, bool stop_on_any, bool done_on_refuted = true, uint32_t active = 0, int auto_slice_ms = 20) {
    if (active == 0 || active > s.n_alloc) active = s.n_alloc;   // workers [0, active) run this slice
    MsParams prm{};
    prm.n_workers = active;
    const bool deterministic = s.opts.deterministic > 0;
    prm.slice_conflicts = s.opts.slice_conflicts > 0 ? (uint32_t)s.opts.slice_conflicts : (deterministic ? 200u : 0xffffffffu);
    prm.slice_props = 0;
    // default: time-bounded slices (all workers stop together; no straggler tail), 20 ms
    const int slice_ms = deterministic ? 0 : (s.opts.slice_ms > 0 ? s.opts.slice_ms : (s.opts.slice_conflicts > 0 ? 0 : auto_slice_ms));
    prm.slice_ticks = slice_ms > 0 ? (uint64_t)slice_ms * 100000ull : 0;
    prm.stop_flag = s.stop_flag;
    prm.stop_on_any = stop_on_any && !deterministic ? 1 : 0;
    prm.max_groups = s.opts.max_groups > 0 ? s.opts.max_groups : MS_MAX_GROUPS;
    prm.any_done = s.d_any_done.p;
    prm.done_on_refuted = done_on_refuted ? 1 : 0;
    prm.proof_buf = s.d_proof.p;
    prm.proof_len = s.d_proof_len.p;
    prm.proof_cap = s.proof_cap;
    prm.reduce_first = s.opts.reduce_first > 0 ? (uint32_t)s.opts.reduce_first : 2000u;
    prm.reduce_inc = s.opts.reduce_inc > 0 ? (uint32_t)s.opts.reduce_inc : 300u;
    prm.rephase = s.opts.rephase;
    prm.restart_k_pct = s.opts.restart_k_pct;
    prm.restart_k2_pct = s.opts.restart_k2_pct;
    prm.import_pct = s.opts.import_pct;
    prm.vivify = s.opts.vivify;
    const bool share = mode == 0 && s.share_slots != 0;
    const uint32_t share_intake_cap = (uint32_t)std::max(16, slice_ms > 0 ? 16 * slice_ms : 256);   // 16 clauses per ms of slice
    if (share) {
        prm.share_pool = s.d_share_pool.p;
        prm.share_n = s.d_share_n.p;
        prm.share_slots = s.share_slots;
        prm.share_max_lbd = s.opts.share_lbd > 0 ? (uint32_t)s.opts.share_lbd : 4u;
        prm.share_max_len = s.opts.share_len > 0 ? (uint32_t)s.opts.share_len : (uint32_t)MS_SHARE_MAXLEN;
        prm.share_interval = s.opts.share_interval > 0 ? (uint32_t)s.opts.share_interval : 0xffffffffu;
    }
    // Assignment (2 bits / variable) and analysis marks (1 bit) in LDS when this launch's workers per CU leave room
    // (160 KB per CU; a workgroup's static 3 KB aside): 16 workers per CU -> 9 KB each (the round-1 rule), one per CU
    // -> up to 64 KB, which covers rect 64x64.  State is written back to HBM at every slice end, so consecutive
    // launches may differ.
    bool lds = s.lds_val;
    if (mode == 0 && s.opts.lds_val == 0) {
        const uint32_t per_cu = (active + 255) / 256;
        lds = s.lds_val_bytes <= std::min<uint32_t>(64 * 1024, 150 * 1024 / per_cu - (per_cu <= 8 ? 14 : 6) * 1024);   // (a workgroup's static LDS aside: 5.2 KB, 13.2 KB in the builds with the sort buffer)
    }
    const uint32_t dyn = lds ? s.lds_val_bytes : 0;
    HIPCHK(hipEventRecord(s.ev0, s.stream));
    if (mode == 0) {
        // the build compiled for the launch's waves per SIMD: 1 (<= 1024 workers: the SIMD's whole register file, everything
        // inlined), 2 (<= 2048: no spills either), else the full fleet's
        int wps = s.opts.one_per_simd < 0 ? MS_SEARCH_WAVES_PER_SIMD : (active <= 1024 ? 1 : (active <= 2048 ? 2 : MS_SEARCH_WAVES_PER_SIMD));
        if (s.opts.one_per_simd == 2 || s.opts.one_per_simd == 4) wps = std::max(wps, s.opts.one_per_simd == 2 ? 2 : MS_SEARCH_WAVES_PER_SIMD);   // (A/B: a build for more waves)
#define MS_LAUNCH_SEARCH(LVV, W) hipLaunchKernelGGL((ms_search_kernel<LVV, W>), dim3(active), dim3(MS_WAVE), (LVV) ? dyn : 0, s.stream, s.sh, s.L, s.d_slabs.p, prm)
        if (lds) {
            if (wps == 1) MS_LAUNCH_SEARCH(true, 1);
            else if (wps == 2) MS_LAUNCH_SEARCH(true, 2);
            else MS_LAUNCH_SEARCH(true, MS_SEARCH_WAVES_PER_SIMD);
        } else {
            if (wps == 1) MS_LAUNCH_SEARCH(false, 1);
            else if (wps == 2) MS_LAUNCH_SEARCH(false, 2);
            else MS_LAUNCH_SEARCH(false, MS_SEARCH_WAVES_PER_SIMD);
        }
#undef MS_LAUNCH_SEARCH
    } else if (mode == 2) {
        if (lds) hipLaunchKernelGGL(ms_probe_kernel<true>, dim3(active), dim3(MS_WAVE), dyn, s.stream, s.sh, s.L, s.d_slabs.p, prm);
        else hipLaunchKernelGGL(ms_probe_kernel<false>, dim3(active), dim3(MS_WAVE), 0, s.stream, s.sh, s.L, s.d_slabs.p, prm);
    } else {
        if (lds) hipLaunchKernelGGL(ms_bcp_kernel<true>, dim3(active), dim3(MS_WAVE), dyn, s.stream, s.sh, s.L, s.d_slabs.p, prm);
        else hipLaunchKernelGGL(ms_bcp_kernel<false>, dim3(active), dim3(MS_WAVE), 0, s.stream, s.sh, s.L, s.d_slabs.p, prm);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s.ev1, s.stream));
    if (share) {
        if ((++s.share_slices & 255) == 0)   // forget old signatures before the set fills up (a clause may then be passed on twice)
            HIPCHK(hipMemsetAsync(s.d_share_hash.p, 0, sizeof(unsigned long long) * s.share_hash_n, s.stream));
        HIPCHK(hipMemsetAsync(s.d_share_intake.p, 0, sizeof(uint32_t), s.stream));
        const bool ordered = s.opts.deterministic > 0;
        hipLaunchKernelGGL(ms_share_collect_kernel, dim3(ordered ? 1 : (s.n_alloc + 63) / 64), dim3(ordered ? 1 : 64), 0, s.stream, s.L, s.d_slabs.p,
                           s.n_alloc, s.d_share_pool.p, s.share_slots, s.d_share_n.p, s.d_share_hash.p, s.share_hash_n - 1,
                           s.d_share_intake.p, share_intake_cap, ordered ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventSynchronize(s.ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s.ev0, s.ev1));
    s.stats.kernel_seconds += ms * 1e-3;
    s.stats.kernel_launches++;
    return SliceResult{ms};
}

void launch_probe(mi355sat& s) { launch_slice(s, 2, false); }

// DRUP text (DIMACS literals, one lemma per line).  Order: what the simplification derived, then after every
// slice the clauses each worker learnt in it (worker by worker, each in its own derivation order), finally the empty
// clause.  That order makes every line a RUP consequence of the lines before it: a learnt clause depends on its
// worker's earlier clauses and on exchanged clauses, and the exchange only hands on clauses of EARLIER slices.
void proof_open(mi355sat& s) {
    s.proof_file = fopen(s.proof_path.c_str(), "w");
    if (!s.proof_file) throw HipErr{"cannot open proof file " + s.proof_path};
    for (int32_t l : s.simp_proof) {   // (caller's numbering already)
        if (l < 0) fputs("0\n", s.proof_file);
        else fprintf(s.proof_file, "%d ", (l & 1) ? -((l >> 1) + 1) : ((l >> 1) + 1));
    }
}
void proof_drain(mi355sat& s) {
    if (!s.proof_file || !s.d_proof_len.p) return;
    const uint32_t W = (uint32_t)s.d_proof_len.n;
    std::vector<uint32_t> len(W);
    HIPCHK(hipMemcpy(len.data(), s.d_proof_len.p, sizeof(uint32_t) * W, hipMemcpyDeviceToHost));
    std::vector<uint32_t> inv(s.perm.size());
    for (uint32_t e = 0; e < s.perm.size(); e++) inv[s.perm[e]] = e;
    std::vector<int32_t> buf;
    bool any = false;
    for (uint32_t w = 0; w < W; w++) {
        if (!len[w]) continue;
        any = true;
        if (len[w] > s.proof_cap) throw HipErr{"proof buffer overflow (a worker learnt more in one slice than its log holds)"};
        buf.resize(len[w]);
        HIPCHK(hipMemcpy(buf.data(), s.d_proof.p + (size_t)w * s.proof_cap, sizeof(int32_t) * len[w], hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < len[w]; i++) {
            if (buf[i] == -2) fputs("d ", s.proof_file);       // deletion line
            else if (buf[i] < 0) fputs("0\n", s.proof_file);
            else fprintf(s.proof_file, "%d ", (buf[i] & 1) ? -((int)inv[buf[i] >> 1] + 1) : ((int)inv[buf[i] >> 1] + 1));
        }
    }
    if (any) { HIPCHK(hipMemsetAsync(s.d_proof_len.p, 0, sizeof(uint32_t) * W, s.stream)); HIPCHK(hipStreamSynchronize(s.stream)); }
}
void proof_close(mi355sat& s, bool unsat) {
    if (!s.proof_file) return;
    if (unsat) fputs("0\n", s.proof_file);
    fclose(s.proof_file);
    s.proof_file = nullptr;
}

const char* status_text(int st) {
    switch (st) {
        case MS_ST_ERR_POOL: return "device watch pool exhausted";
        case MS_ST_ERR_LEARNT: return "device learnt-clause store exhausted";
        case MS_ST_ERR_INTERNAL: return "device solver internal error";
        default: return "unknown device status";
    }
}

// Shared driver for solve(), solve_batch() and the stepwise sweep API.
// Worker w starts on instance w % n_instances; workers of decided (or withdrawn) instances move on
// to the open ones (w_inst).
struct Sweep {
    uint32_t n_instances = 0;
    std::vector<int32_t> results, winner;
    std::vector<uint8_t> dropped;                // withdrawn by the caller: result stays INTERRUPTED, counts as decided
    std::vector<double> weights;                 // share of the fleet each open instance should get (empty: equal)
    bool weights_dirty = false;
    std::vector<int32_t> base_assump;            // internal literals
    std::vector<uint64_t> base_off;
    uint64_t n_moved = 0;
    float ramp_ms = 0;                           // kernel time of this sweep so far
    uint32_t decided = 0;
    bool stop_at_first = false;
    bool active = false;
    std::vector<MsState> sts;
    uint64_t conflicts = 0;
    // cube scheduler (opts.cube_split): every busy worker owns one cube = its instance's base
    // assumptions + split literals; the cubes of an instance partition its search space
    bool split = false;
    std::vector<int32_t> w_inst;
    std::vector<std::vector<int32_t>> w_cube;   // internal literals
    std::vector<uint8_t> w_busy;
    std::vector<uint64_t> w_conf0;               // worker's conflict count when it got its cube
    std::vector<uint32_t> open;                  // open cubes per instance
    uint64_t n_splits = 0, n_closed = 0;
    DevBuf<int32_t> d_upd, d_data;
};

int sweep_begin(mi355sat& s, Sweep& sw, const std::vector<int32_t>& assump, const std::vector<uint64_t>& assump_off,
                uint32_t n_instances, bool stop_at_first) {
    Prepared P;
    {
        Formula F;
        normalise(s, F);
        F.log_proof = !s.proof_path.empty();
        for (int32_t d : assump)
            if (d != 0 && (uint64_t)(d < 0 ? -(int64_t)d : d) <= F.nv) F.frozen_lits.push_back(to_internal(d));
        simplify_formula(s, F);
        build_csr(s, F, /*units_propagated=*/true, P);
        s.subst = F.subst;
        s.elims.swap(F.elims);
        s.elim_lits.swap(F.elim_lits);
        s.stats.simp_eliminated = F.n_eliminated;
        s.simp_proof.swap(F.proof);
        s.stats.simp_units = F.n_failed + F.n_necessary;
        s.stats.simp_equivalences = F.n_equiv;
        s.stats.simp_clauses_removed = F.n_subsumed + F.n_strengthened;
    }
    if (!s.proof_path.empty()) {
        if (n_instances != 1) throw HipErr{"a proof can only be logged for a plain solve()"};
        proof_open(s);
    }
    sw.n_instances = n_instances;
    sw.stop_at_first = stop_at_first;
    sw.decided = 0;
    sw.active = false;
    sw.results.assign(n_instances, MI355SAT_INTERRUPTED);
    sw.winner.assign(n_instances, -1);
    sw.dropped.assign(n_instances, 0);
    sw.n_moved = 0;
    sw.ramp_ms = 0;
    if (P.unsat) {
        std::fill(sw.results.begin(), sw.results.end(), MI355SAT_UNSAT);
        sw.decided = n_instances;
        s.trivially_unsat = true;
        return 0;
    }
    // default fleet: the whole GPU (16 waves per CU) for large formulas; mid-size ones measured fastest to a verdict
    // with 1024 workers (rect 24x24 ladder), small ones do not pay for more than one worker per CU
    uint32_t want = s.opts.workers > 0 ? (uint32_t)s.opts.workers
                                       : (s.offs.size() > 100000 ? MS_SEARCH_WAVES_PER_SIMD * 1024u : (s.offs.size() > 20000 ? 1024u : 256u));
    if (want < n_instances) want = n_instances;
    want = want / n_instances * n_instances;
    s.d_proof.release();
    s.d_proof_len.release();
    s.proof_cap = 0;
    sw.split = s.opts.cube_split > 0 && want > n_instances;   // opt-in: see DESIGN.md (measured: not yet a win)
    std::vector<int32_t> a_int(assump.size());
    for (size_t i = 0; i < assump.size(); i++) {
        int32_t d = assump[i];
        if (d == 0 || (uint64_t)(d < 0 ? -(int64_t)d : d) > P.n_vars) throw HipErr{"assumption literal out of range"};
        int32_t l = to_internal(d);
        while (s.subst[l >> 1] != 2 * (l >> 1)) l = s.subst[l >> 1] ^ (l & 1);     // a variable replaced by an equivalent literal
        a_int[i] = 2 * (int32_t)P.perm[l >> 1] | (l & 1);
    }
    uint32_t max_assumps = 0;
    for (uint32_t i = 0; i < n_instances; i++)
        max_assumps = std::max<uint32_t>(max_assumps, (uint32_t)(assump_off[i + 1] - assump_off[i]));
    const uint32_t assump_cap = sw.split ? max_assumps + 512 : max_assumps;
    const uint32_t initial = (s.opts.ramp >= 0 && !sw.split && s.opts.deterministic <= 0) ? std::max(256u, n_instances) / n_instances * n_instances : 0;
    upload_formula(s, P, assump_cap, 0, want, initial);
    if (s.n_workers < n_instances) throw HipErr{"not enough device memory for one worker per instance"};
    s.n_workers = s.n_workers / n_instances * n_instances;
    s.n_alloc = std::min(s.n_alloc, s.n_workers);
    if (!s.proof_path.empty()) {   // one log per worker, drained after every slice: a slice's lemmas (~10^4 learnt clauses) plus the
        // deletion lines of one clause-database reduction, which may drop half of a full learnt store at once (lemmas that do
        // not fit fail the solve - the proof would be wrong; deletion lines that do not fit are dropped - they are optional)
        s.proof_cap = (uint32_t)std::min<uint64_t>(1u << 23, (1u << 20) + s.L.learnt_lit_cap / 2 + 2ull * s.L.learnt_cap);
        s.d_proof.alloc((size_t)s.n_workers * s.proof_cap);
        s.d_proof_len.alloc(s.n_workers);
        HIPCHK(hipMemsetAsync(s.d_proof_len.p, 0, sizeof(uint32_t) * s.n_workers, s.stream));
    }
    reset_workers(s);
    customize(s, &a_int, &assump_off, nullptr, nullptr, n_instances, sw.split ? (int32_t)n_instances : -1);
    HIPCHK(hipStreamSynchronize(s.stream));
    const uint32_t W = s.n_workers;
    s.stats.workers = W;
    sw.base_assump = a_int;
    sw.base_off = assump_off;
    sw.w_inst.assign(W, 0);
    sw.w_cube.assign(W, {});
    sw.w_busy.assign(W, 0);
    sw.w_conf0.assign(W, 0);
    sw.open.assign(n_instances, 0);
    for (uint32_t w = 0; w < W; w++) {
        uint32_t inst = w % n_instances;
        sw.w_inst[w] = (int32_t)inst;
        if (!sw.split || w < n_instances) {
            sw.w_busy[w] = 1;
            sw.w_cube[w].assign(a_int.begin() + assump_off[inst], a_int.begin() + assump_off[inst + 1]);
            sw.open[inst]++;
        }
    }
    sw.active = true;
    return 0;
}

// Work stealing between two slices: idle workers take over sub-cubes split off the oldest free
// decisions of running workers.  Victim with cube C and decisions d1..dm keeps C,d1..dm; thief j
// gets C,d1..d(j-1),~dj — together they partition C, so an instance is UNSAT exactly when all its
// cubes are closed.
void schedule_cubes(mi355sat& s, Sweep& sw) {
    const uint32_t W = s.n_workers, cap = s.L.assump_cap;
    std::vector<int32_t> upd, data;
    auto push_update = [&](uint32_t w, int32_t status, int32_t restart) {
        upd.insert(upd.end(), {(int32_t)w, status, restart, (int32_t)sw.w_cube[w].size(), (int32_t)data.size()});
        data.insert(data.end(), sw.w_cube[w].begin(), sw.w_cube[w].end());
    };
    std::vector<uint32_t> idle, victims;
    for (uint32_t w = 0; w < W; w++) {
        const bool undecided = sw.results[sw.w_inst[w]] == MI355SAT_INTERRUPTED;
        if (sw.w_busy[w] && !undecided) {   // its instance was decided by someone else: park it
            sw.w_busy[w] = 0;
            if (sw.sts[w].status == MS_ST_RUNNING) { sw.w_cube[w].clear(); push_update(w, MS_ST_PARKED, 0); sw.sts[w].status = MS_ST_PARKED; }
        }
        if (!sw.w_busy[w]) idle.push_back(w);
        else if (sw.sts[w].status == MS_ST_RUNNING && sw.sts[w].n_split > 0) victims.push_back(w);
    }
    // hardest cubes first: most conflicts spent on the current cube
    std::sort(victims.begin(), victims.end(), [&](uint32_t a, uint32_t b) {
        return sw.sts[a].conflicts - sw.w_conf0[a] > sw.sts[b].conflicts - sw.w_conf0[b];
    });
    std::vector<uint32_t> taken(W, 0);
    size_t next_idle = 0;
    for (uint32_t round = 0; round < MS_SPLIT_MAX && next_idle < idle.size(); round++) {
        bool any = false;
        for (uint32_t v : victims) {
            if (next_idle >= idle.size()) break;
            if (taken[v] != round || (int32_t)round >= sw.sts[v].n_split) continue;
            if (sw.w_cube[v].size() + 1 > cap) continue;
            const int32_t d = sw.sts[v].split[round];
            const uint32_t t = idle[next_idle++];
            sw.w_cube[t] = sw.w_cube[v];
            sw.w_cube[t].push_back(d ^ 1);
            sw.w_cube[v].push_back(d);
            sw.w_inst[t] = sw.w_inst[v];
            sw.w_busy[t] = 1;
            sw.w_conf0[t] = sw.sts[t].conflicts;
            sw.open[sw.w_inst[v]]++;
            sw.n_splits++;
            taken[v] = round + 1;
            push_update(t, MS_ST_RUNNING, 1);
            any = true;
        }
        if (!any) break;
    }
    for (uint32_t v : victims)
        if (taken[v]) push_update(v, MS_ST_RUNNING, 0);
    if (upd.empty()) return;
    sw.d_upd.upload(upd, s.stream);
    sw.d_data.upload(data.empty() ? std::vector<int32_t>{0} : data, s.stream);
    hipLaunchKernelGGL(ms_assign_kernel, dim3((uint32_t)(upd.size() / 5)), dim3(64), 0, s.stream, s.L, s.d_slabs.p,
                       (uint32_t)(upd.size() / 5), sw.d_upd.p, sw.d_data.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s.stream));
}

inline bool inst_open(const Sweep& sw, uint32_t inst) { return sw.results[inst] == MI355SAT_INTERRUPTED && !sw.dropped[inst]; }

// Portfolio mode, between two slices: the workers of instances that are decided (or withdrawn) move to
// the open instances with the fewest workers.  They keep their learnt clauses (consequences of the
// formula alone) and only swap their assumption list; the worker holding a SAT instance's model stays.
void rebalance_workers(mi355sat& s, Sweep& sw) {
    const uint32_t n_instances = sw.n_instances;
    std::vector<uint32_t> cnt(n_instances, 0), movable;
    for (uint32_t w = 0; w < s.n_alloc; w++) {   // (a worker without a slab yet is picked up after grow_workers)
        const uint32_t inst = (uint32_t)sw.w_inst[w];
        const int st = sw.sts[w].status;
        if (inst_open(sw, inst)) { if (st == MS_ST_RUNNING) cnt[inst]++; continue; }
        if (sw.winner[inst] == (int32_t)w && sw.results[inst] == MI355SAT_SAT) continue;   // keeps the model
        if (st == MS_ST_RUNNING || st == MS_ST_SAT || st == MS_ST_REFUTED || st == MS_ST_PARKED) movable.push_back(w);
    }
    std::vector<uint32_t> open;
    for (uint32_t i = 0; i < n_instances; i++) if (inst_open(sw, i)) open.push_back(i);
    std::vector<int32_t> upd, data;
    auto park = [&](uint32_t w) {
        if (sw.sts[w].status != MS_ST_RUNNING) return;
        upd.insert(upd.end(), {(int32_t)w, MS_ST_PARKED, 0, 0, (int32_t)data.size()});
        sw.sts[w].status = MS_ST_PARKED;
        sw.w_busy[w] = 0;
    };
    if (open.empty() || s.opts.rebalance < 0) {
        for (uint32_t w : movable) park(w);
    } else {
        // Each open instance's share of the fleet follows its weight (mi355sat_sweep_set_weights; default equal).
        // Workers of instances well above their share (> 25 % and > 2 workers) are taken off them too - highest
        // worker index first, never the last one - so that a caller's change of priorities takes effect.
        double wsum = 0;
        for (uint32_t i : open) wsum += sw.weights.empty() ? 1.0 : std::max(1e-6, sw.weights[i]);
        uint32_t total = (uint32_t)movable.size();
        for (uint32_t i : open) total += cnt[i];
        std::vector<double> target(n_instances, 0);
        for (uint32_t i : open) target[i] = std::max(1.0, total * (sw.weights.empty() ? 1.0 : std::max(1e-6, sw.weights[i])) / wsum);
        if (!sw.weights.empty()) {
            std::vector<uint32_t> surplus(n_instances, 0);
            for (uint32_t i : open)
                if (cnt[i] > target[i] * 1.25 + 2) surplus[i] = cnt[i] - (uint32_t)target[i];
            for (uint32_t w = s.n_alloc; w-- > 0;) {
                const uint32_t inst = (uint32_t)sw.w_inst[w];
                if (!inst_open(sw, inst) || !surplus[inst] || sw.sts[w].status != MS_ST_RUNNING || cnt[inst] <= 1) continue;
                surplus[inst]--;
                cnt[inst]--;
                movable.push_back(w);
            }
        }
        // an open instance nobody works on (reopened with every worker busy elsewhere, or its workers' slabs came later)
        // takes one worker from the instance that has the most - weights or not: it would never be decided otherwise
        for (uint32_t i : open) {
            if (cnt[i] > 0 || !movable.empty()) continue;
            uint32_t rich = open[0];
            for (uint32_t j : open) if (cnt[j] > cnt[rich]) rich = j;
            if (cnt[rich] <= 1) break;
            for (uint32_t w = s.n_alloc; w-- > 0;)
                if ((uint32_t)sw.w_inst[w] == rich && sw.sts[w].status == MS_ST_RUNNING) { movable.push_back(w); cnt[rich]--; break; }
        }
        for (uint32_t w : movable) {
            uint32_t best = open[0];
            double best_need = -1e30;
            for (uint32_t i : open) {   // the instance furthest below its share, relative to it
                const double need = (target[i] - cnt[i]) / target[i];
                if (need > best_need) { best_need = need; best = i; }
            }
            cnt[best]++;
            sw.w_inst[w] = (int32_t)best;
            sw.w_busy[w] = 1;
            sw.w_cube[w].assign(sw.base_assump.begin() + sw.base_off[best], sw.base_assump.begin() + sw.base_off[best + 1]);
            upd.insert(upd.end(), {(int32_t)w, MS_ST_RUNNING, 1, (int32_t)sw.w_cube[w].size(), (int32_t)data.size()});
            data.insert(data.end(), sw.w_cube[w].begin(), sw.w_cube[w].end());
            sw.sts[w].status = MS_ST_RUNNING;
            sw.n_moved++;
        }
    }
    if (upd.empty()) return;
    sw.d_upd.upload(upd, s.stream);
    sw.d_data.upload(data.empty() ? std::vector<int32_t>{0} : data, s.stream);
    hipLaunchKernelGGL(ms_assign_kernel, dim3((uint32_t)(upd.size() / 5)), dim3(64), 0, s.stream, s.L, s.d_slabs.p,
                       (uint32_t)(upd.size() / 5), sw.d_upd.p, sw.d_data.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s.stream));
}

// One slice of the search kernel over all workers.  Returns 0 or a negative error.
int sweep_step(mi355sat& s, Sweep& sw) {
    if (!sw.active) return 0;
    const uint32_t n_instances = sw.n_instances;
    // Ramp-up: a worker alone on its CU runs ~3x faster than one of 16, and an easy instance is decided by ONE
    // worker's few hundred conflicts - so the first 100 ms of kernel time go to one worker per CU, the next
    // 300 ms to four, and only a search that is still open after that gets the whole fleet.  (Measured: rect
    // 32x32 k=120 0.167 -> 0.088 s; 250 / 1000 ms thresholds gain nothing more at 64x64 - a worker there is bound
    // by DRAM latency even when alone - and delay the rect 24x24 ladder by 0.5-1 s.)
    uint32_t active = s.n_workers;
    if (s.opts.ramp >= 0 && !sw.split && s.opts.deterministic <= 0) {
        const uint32_t want = sw.ramp_ms < 100.f ? 256u : (sw.ramp_ms < 400.f ? 1024u : s.n_workers);
        active = std::min(s.n_workers, std::max(want, n_instances) / n_instances * n_instances);
    }
    if (active > s.n_alloc) {
        const uint32_t had = s.n_alloc;
        grow_workers(s, n_instances, active);
        // the new workers start on instance w % n_instances - which may be decided or withdrawn by now: move them before the slice
        if (n_instances > 1 && !sw.split && s.n_alloc > had && !sw.sts.empty() && (sw.decided > 0 || std::count(sw.dropped.begin(), sw.dropped.end(), 1) > 0)) {
            sw.sts.resize(s.n_workers, MsState{});
            rebalance_workers(s, sw);
        }
    }
    // default slice length: 20 ms while a solve is young (easy bounds are decided within a few), 50 ms after one second
    // and 100 ms after ten of kernel time - the host's share per slice (collecting states, the caller's loop) was a
    // quarter of the wall-clock of the rect 26x26 ladder with 10 ms slices
    const int auto_ms = sw.ramp_ms < 1000.f ? 20 : (sw.ramp_ms < 10000.f ? 50 : 100);
    SliceResult sr = launch_slice(s, 0, /*stop_on_any=*/n_instances == 1 || sw.stop_at_first, /*done_on_refuted=*/!sw.split, active, auto_ms);
    sw.ramp_ms += sr.ms;
    proof_drain(s);
    gather_states(s, sw.sts);
    int rc = 0;
    uint64_t confl = 0;
    for (uint32_t w = 0; w < s.n_workers; w++) {
        const MsState& st = sw.sts[w];
        confl += st.conflicts;
        if (st.status < 0) {
            set_error(&s, status_text(st.status));
            rc = st.status == MS_ST_ERR_INTERNAL ? MI355SAT_ERR_STATE : MI355SAT_ERR_OOM;
        }
        if (!sw.w_busy[w]) continue;
        const uint32_t inst = (uint32_t)sw.w_inst[w];
        if (st.status == MS_ST_REFUTED || st.status == MS_ST_UNSAT) {   // this worker's cube is closed
            sw.w_busy[w] = 0;
            sw.open[inst]--;
            sw.n_closed++;
        }
        if (st.status == MS_ST_UNSAT) {   // refuted without any decision: the formula itself, whatever the assumptions
            for (uint32_t i = 0; i < n_instances; i++)
                if (inst_open(sw, i)) { sw.results[i] = MI355SAT_UNSAT; sw.winner[i] = (int32_t)w; sw.decided++; }
            continue;
        }
        if (!inst_open(sw, inst)) continue;
        if (st.status == MS_ST_SAT) { sw.results[inst] = MI355SAT_SAT; sw.winner[inst] = (int32_t)w; sw.decided++; }
        else if (st.status == MS_ST_UNSAT || (st.status == MS_ST_REFUTED && (!sw.split || sw.open[inst] == 0))) {
            // the formula itself refuted, or (with splitting) the last open cube of the instance closed;
            // without splitting every worker holds the instance's whole search space
            sw.results[inst] = MI355SAT_UNSAT; sw.winner[inst] = (int32_t)w; sw.decided++;
        }
    }
    sw.conflicts = confl;
    if (s.opts.verbose) {
        uint64_t props = 0, nl = 0, busy = 0, viv = 0, vivl = 0;
        for (auto& st : sw.sts) { props += st.propagations; nl += st.n_learnts; viv += st.n_vivified; vivl += st.n_viv_lits; }
        if (viv) fprintf(stderr, "[mi355sat] vivified %llu clauses, %llu literals removed\n", (unsigned long long)viv, (unsigned long long)vivl);
        for (auto b : sw.w_busy) busy += b;
        fprintf(stderr, "[mi355sat] slice: decided %u/%u conflicts=%llu props=%llu kernel=%.3fs busy=%llu/%u splits=%llu closed=%llu kept=%llu\n",
                sw.decided, n_instances, (unsigned long long)confl, (unsigned long long)props, s.stats.kernel_seconds,
                (unsigned long long)busy, s.n_workers, (unsigned long long)sw.n_splits, (unsigned long long)sw.n_closed,
                (unsigned long long)nl);
    }
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(s.d_any_done.p, 0, sizeof(int32_t), s.stream));
    if (sw.decided == n_instances || (sw.stop_at_first && sw.decided > 0)) return 0;
    if (sw.split) schedule_cubes(s, sw);
    else if (n_instances > 1 && (sw.decided > 0 || sw.weights_dirty)) rebalance_workers(s, sw);
    sw.weights_dirty = false;
    return 0;
}

bool sweep_finished(const mi355sat& s, const Sweep& sw) {
    if (!sw.active) return true;
    if (sw.decided == sw.n_instances || (sw.stop_at_first && sw.decided > 0)) return true;
    if (s.interrupted.load() || *s.stop_flag) return true;
    if (s.opts.conflict_budget > 0 && (int64_t)sw.conflicts >= s.opts.conflict_budget) return true;
    return false;
}

// An interrupt is consumed by the solve it stops (or, if it came before solve(), by the next one, which
// returns INTERRUPTED at once): later solves on the same handle run normally.
void consume_interrupt(mi355sat& s) {
    if (s.interrupted.exchange(0)) __atomic_store_n(s.stop_flag, 0, __ATOMIC_SEQ_CST);
}

void sweep_end(mi355sat& s, Sweep& sw) {
    if (sw.active) accumulate_stats(s, sw.sts);
    sw.active = false;
    consume_interrupt(s);
}

int run_search(mi355sat& s, const std::vector<int32_t>& assump, const std::vector<uint64_t>& assump_off,
               uint32_t n_instances, std::vector<int32_t>& results, std::vector<int32_t>& winner, bool stop_at_first) {
    Sweep sw;
    int rc = sweep_begin(s, sw, assump, assump_off, n_instances, stop_at_first);
    while (!rc && !sweep_finished(s, sw)) rc = sweep_step(s, sw);
    sweep_end(s, sw);
    results = sw.results;
    winner = sw.winner;
    return rc;
}

}  // namespace

struct SweepHolder { Sweep sw; mi355sat_stats_t base; };

// =============================================================================== C ABI
extern "C" {

uint64_t mi355sat_abi_sizes(uint64_t* stats_size) {
    if (stats_size) *stats_size = sizeof(mi355sat_stats_t);
    return sizeof(mi355sat_opts);
}

void mi355sat_release_cached_memory(void) {
    std::lock_guard<std::mutex> g(g_slab_cache.mu);
    for (int d = 0; d < 64; d++)
        if (g_slab_cache.p[d]) {
            if (hipSetDevice(d) == hipSuccess) (void)hipFree(g_slab_cache.p[d]);
            g_slab_cache.p[d] = nullptr;
            g_slab_cache.bytes[d] = 0;
        }
}

const char* mi355sat_signature(void) { return "mi355sat 0.1 (HIP/gfx950 wave-parallel CDCL)"; }

const char* mi355sat_last_error(const mi355sat* s) {
    if (s) return s->err.c_str();
    std::lock_guard<std::mutex> g(g_new_error_mu);
    return g_new_error.c_str();
}

mi355sat* mi355sat_new(const mi355sat_opts* opts) {
    auto fail = [](const std::string& m) -> mi355sat* {
        std::lock_guard<std::mutex> g(g_new_error_mu);
        g_new_error = m;
        return nullptr;
    };
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(std::string("no usable HIP device (mi355sat has no CPU fallback): ") +
                    (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    mi355sat* s = new (std::nothrow) mi355sat;
    if (!s) return fail("out of host memory");
    if (opts) s->opts = *opts;
    int dev = 0;
    if (opts && opts->device >= 0) dev = opts->device;
    else if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= n) { delete s; return fail("device ordinal out of range"); }
    s->device = dev;
    try {
        HIPCHK(hipSetDevice(dev));
        HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&s->ev0));
        HIPCHK(hipEventCreate(&s->ev1));
        HIPCHK(hipHostMalloc((void**)&s->stop_flag, sizeof(int32_t), hipHostMallocMapped));
        *s->stop_flag = 0;
    } catch (HipErr& he) {
        std::string m = he.msg;
        delete s;
        return fail(m);
    }
    return s;
}

void mi355sat_free(mi355sat* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    delete s->sweep;
    if (s->proof_file) fclose(s->proof_file);
    s->d_cl_lits.release(); s->d_bin_lits.release();
    s->d_tern_pairs.release(); s->d_tern_owner.release();
    s->d_template.release(); s->d_slabs.release(); s->d_states.release(); s->d_any_done.release();
    s->d_proof.release(); s->d_proof_len.release();
    s->d_assump.release(); s->d_script.release(); s->d_assump_off.release(); s->d_script_off.release();
    if (s->stop_flag) (void)hipHostFree(s->stop_flag);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int mi355sat_reserve(mi355sat* s, uint64_t n_vars) {
    if (!s) return MI355SAT_ERR_ARG;
    if (n_vars > s->max_var) s->max_var = n_vars;
    s->stats.max_var = s->max_var;
    return 0;
}

static int add_clause_impl(mi355sat* s, const int32_t* l, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) {
        int32_t d = l[i];
        if (d == 0 || d == INT32_MIN) { s->err = "literal 0 inside a clause"; return MI355SAT_ERR_ARG; }
        uint64_t v = (uint64_t)(d < 0 ? -(int64_t)d : d);
        if (v > MS_MAX_VARS) { s->err = "variable index too large"; return MI355SAT_ERR_ARG; }
        if (v > s->max_var) s->max_var = v;
    }
    s->lits.insert(s->lits.end(), l, l + n);
    s->offs.push_back(s->lits.size());
    s->stats.n_clauses++;
    s->stats.max_var = s->max_var;
    s->stats.avg_clause_len = (double)s->lits.size() / (double)s->stats.n_clauses;
    return 0;
}

int mi355sat_add_cnf(mi355sat* s, const int32_t* lits, const uint64_t* offsets, uint64_t n_clauses) {
    if (!s || (n_clauses && (!offsets || (!lits && offsets[n_clauses] > offsets[0])))) return MI355SAT_ERR_ARG;
    try {
        for (uint64_t c = 0; c < n_clauses; c++) {
            if (offsets[c + 1] < offsets[c]) { s->err = "offsets not monotone"; return MI355SAT_ERR_ARG; }
            int rc = add_clause_impl(s, lits + offsets[c], offsets[c + 1] - offsets[c]);
            if (rc) return rc;
        }
    } catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
    return 0;
}

int mi355sat_add(mi355sat* s, int32_t lit_or_0) {
    if (!s) return MI355SAT_ERR_ARG;
    try {
        if (lit_or_0 != 0) { s->pending.push_back(lit_or_0); return 0; }
        int rc = add_clause_impl(s, s->pending.data(), s->pending.size());
        s->pending.clear();
        return rc;
    } catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
}

void mi355sat_interrupt(mi355sat* s) {
    if (!s) return;
    s->interrupted.store(1);
    if (s->stop_flag) __atomic_store_n(s->stop_flag, 1, __ATOMIC_SEQ_CST);
}

int mi355sat_set_proof_path(mi355sat* s, const char* path) {
    if (!s) return MI355SAT_ERR_ARG;
    s->proof_path = path ? path : "";
    return 0;
}

int mi355sat_solve(mi355sat* s) {
    if (!s) return MI355SAT_ERR_ARG;
    if (!s->pending.empty()) { s->err = "solve() called inside an unterminated clause"; return MI355SAT_ERR_STATE; }
    const double t0 = now_s();
    int result;
    try {
        HIPCHK(hipSetDevice(s->device));
        std::vector<int32_t> assump;
        std::vector<uint64_t> aoff{0, 0};
        std::vector<int32_t> results, winner;
        int rc = run_search(*s, assump, aoff, 1, results, winner, true);
        if (rc) { proof_close(*s, false); s->stats.solve_seconds += now_s() - t0; return rc; }   // (a truncated proof file is closed, not leaked)
        result = results[0];
        proof_close(*s, result == MI355SAT_UNSAT);
        s->model.clear();
        if (result == MI355SAT_SAT) fetch_model(*s, (uint32_t)winner[0], s->model, s->max_var);
    } catch (HipErr& he) {
        proof_close(*s, false);
        s->err = he.msg;
        s->stats.solve_seconds += now_s() - t0;
        return MI355SAT_ERR_HIP;
    } catch (std::bad_alloc&) {
        s->err = "out of host memory";
        return MI355SAT_ERR_OOM;
    }
    if (result == MI355SAT_SAT) s->stats.n_sat++;
    else if (result == MI355SAT_UNSAT) s->stats.n_unsat++;
    else s->stats.n_terminated++;
    s->stats.solve_seconds += now_s() - t0;
    return result;
}

int mi355sat_solve_batch(mi355sat* s, const int32_t* assumps, const uint64_t* assump_offsets, uint64_t n_instances,
                         int32_t* results_out, int stop_at_first) {
    if (!s || !assump_offsets || !results_out || n_instances == 0) return MI355SAT_ERR_ARG;
    const double t0 = now_s();
    try {
        HIPCHK(hipSetDevice(s->device));
        std::vector<uint64_t> aoff(assump_offsets, assump_offsets + n_instances + 1);
        for (auto& o : aoff) o -= assump_offsets[0];
        std::vector<int32_t> assump;
        if (aoff.back()) assump.assign(assumps + assump_offsets[0], assumps + assump_offsets[n_instances]);
        std::vector<int32_t> results, winner;
        int rc = run_search(*s, assump, aoff, (uint32_t)n_instances, results, winner, stop_at_first != 0);
        if (rc) { s->stats.solve_seconds += now_s() - t0; return rc; }
        s->batch_models.assign(n_instances, {});
        for (uint64_t i = 0; i < n_instances; i++) {
            results_out[i] = results[i];
            if (results[i] == MI355SAT_SAT && winner[i] >= 0) fetch_model(*s, (uint32_t)winner[i], s->batch_models[i], s->max_var);
            if (results[i] == MI355SAT_SAT) s->stats.n_sat++;
            else if (results[i] == MI355SAT_UNSAT) s->stats.n_unsat++;
            else s->stats.n_terminated++;
        }
    } catch (HipErr& he) {
        s->err = he.msg;
        s->stats.solve_seconds += now_s() - t0;
        return MI355SAT_ERR_HIP;
    } catch (std::bad_alloc&) {
        s->err = "out of host memory";
        return MI355SAT_ERR_OOM;
    }
    s->stats.solve_seconds += now_s() - t0;
    return 0;
}

int mi355sat_propagate_batch(mi355sat* s, const int32_t* decisions, const uint64_t* decision_offsets,
                             uint64_t n_instances, int8_t* out_values, uint64_t n_vars, int32_t* out_conflict,
                             int32_t* out_trail_len, int32_t repeat) {
    if (!s || !decision_offsets || n_instances == 0) return MI355SAT_ERR_ARG;
    const double t0 = now_s();
    try {
        HIPCHK(hipSetDevice(s->device));
        Prepared P;
        prepare(*s, /*simplify=*/false, P);
        if (P.unsat) {  // contradictory unit clauses: every instance conflicts before any decision
            for (uint64_t i = 0; i < n_instances; i++) {
                if (out_conflict) out_conflict[i] = 1;
                if (out_trail_len) out_trail_len[i] = 0;
            }
            if (out_values) memset(out_values, 0, n_instances * n_vars);
            return 0;
        }
        std::vector<uint64_t> soff(decision_offsets, decision_offsets + n_instances + 1);
        for (auto& o : soff) o -= decision_offsets[0];
        std::vector<int32_t> script(soff.back());
        uint32_t max_script = 0;
        for (uint64_t i = 0; i < n_instances; i++) max_script = std::max<uint32_t>(max_script, (uint32_t)(soff[i + 1] - soff[i]));
        for (uint64_t k = 0; k < soff.back(); k++) {
            int32_t d = decisions[decision_offsets[0] + k];
            if (d == 0 || (uint64_t)(d < 0 ? -(int64_t)d : d) > P.n_vars) throw HipErr{"decision literal out of range"};
            script[k] = to_device(P.perm, d);
        }
        upload_formula(*s, P, 0, max_script, (uint32_t)n_instances);
        if (s->n_workers < n_instances) throw HipErr{"not enough device memory for the batch"};
        std::vector<MsState> sts;
        if (repeat < 1) repeat = 1;
        for (int r = 0; r < repeat; r++) {
            reset_workers(*s);
            customize(*s, nullptr, nullptr, &script, &soff, (uint32_t)n_instances);
            launch_slice(*s, 1, false);
        }
        gather_states(*s, sts);
        for (uint64_t i = 0; i < n_instances; i++) {
            if (sts[i].status < 0) { s->err = status_text(sts[i].status); return MI355SAT_ERR_OOM; }
            if (out_conflict) out_conflict[i] = sts[i].status == MS_ST_UNSAT ? 1 : 0;
            if (out_trail_len) out_trail_len[i] = sts[i].trail_n;
        }
        // counters of the last repeat only, scaled
        {
            std::vector<MsState> one = sts;
            accumulate_stats(*s, one);
            if (repeat > 1) {
                mi355sat_stats_t& o = s->stats;
                uint64_t props = 0, nw = 0, ncl = 0, nm = 0, ne = 0;
                for (auto& st : sts) { props += st.propagations; nw += st.n_watch; ncl += st.n_cl_lit; nm += st.n_move; ne += st.n_enq; }
                uint64_t k = (uint64_t)(repeat - 1);
                o.propagations += props * k; o.n_deq += props * k; o.n_watch += nw * k; o.n_cl_lit += ncl * k;
                o.n_move += nm * k; o.n_enq += ne * k;
            }
        }
        if (out_values) {
            const size_t nw = P.n_vars;
            std::vector<uint8_t> raw(n_instances * nw + 1);
            if (P.n_vars)
                HIPCHK(hipMemcpy2D(raw.data(), nw, s->d_slabs.p + s->L.val, s->L.slab_bytes, nw,
                                   n_instances, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < n_instances; i++)
                for (uint64_t v = 0; v < n_vars; v++) {
                    uint8_t x = v < P.n_vars ? raw[i * nw + P.perm[v]] : MS_ASG_UNDEF;
                    out_values[i * n_vars + v] = x == MS_ASG_TRUE ? 1 : (x == MS_ASG_FALSE ? -1 : 0);
                }
        }
    } catch (HipErr& he) {
        s->err = he.msg;
        s->stats.solve_seconds += now_s() - t0;
        return MI355SAT_ERR_HIP;
    } catch (std::bad_alloc&) {
        s->err = "out of host memory";
        return MI355SAT_ERR_OOM;
    }
    s->stats.solve_seconds += now_s() - t0;
    return 0;
}

int mi355sat_sweep_begin(mi355sat* s, const int32_t* assumps, const uint64_t* assump_offsets, uint64_t n_instances) {
    if (!s || !assump_offsets || n_instances == 0) return MI355SAT_ERR_ARG;
    try {
        HIPCHK(hipSetDevice(s->device));
        std::vector<uint64_t> aoff(assump_offsets, assump_offsets + n_instances + 1);
        for (auto& o : aoff) o -= assump_offsets[0];
        std::vector<int32_t> assump;
        if (aoff.back()) assump.assign(assumps + assump_offsets[0], assumps + assump_offsets[n_instances]);
        delete s->sweep;
        s->sweep = new SweepHolder;
        s->sweep->base = s->stats;
        return sweep_begin(*s, s->sweep->sw, assump, aoff, (uint32_t)n_instances, false);
    } catch (HipErr& he) { s->err = he.msg; return MI355SAT_ERR_HIP; }
    catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
}

int mi355sat_sweep_step(mi355sat* s, int32_t* results_out, uint64_t* n_decided) {
    if (!s || !s->sweep) return MI355SAT_ERR_STATE;
    const double t0 = now_s();
    try {
        HIPCHK(hipSetDevice(s->device));
        Sweep& sw = s->sweep->sw;
        int rc = sweep_step(*s, sw);
        if (results_out) for (uint32_t i = 0; i < sw.n_instances; i++) results_out[i] = sw.results[i];
        if (n_decided) *n_decided = sw.decided;
        // running totals so that stats() is meaningful between steps
        mi355sat_stats_t keep = s->stats;
        s->stats = s->sweep->base;
        s->stats.kernel_seconds = keep.kernel_seconds;
        s->stats.kernel_launches = keep.kernel_launches;
        s->stats.workers = keep.workers;
        s->stats.simp_units = keep.simp_units; s->stats.simp_equivalences = keep.simp_equivalences;
        s->stats.simp_clauses_removed = keep.simp_clauses_removed; s->stats.simp_eliminated = keep.simp_eliminated;
        s->stats.solve_seconds = keep.solve_seconds + (now_s() - t0);
        accumulate_stats(*s, sw.sts);
        return rc;
    } catch (HipErr& he) { s->err = he.msg; return MI355SAT_ERR_HIP; }
    catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
}

int mi355sat_sweep_drop(mi355sat* s, const uint64_t* instances, uint64_t n) {
    if (!s || !s->sweep || (n && !instances)) return MI355SAT_ERR_STATE;
    try {
        HIPCHK(hipSetDevice(s->device));
        Sweep& sw = s->sweep->sw;
        bool any = false;
        for (uint64_t j = 0; j < n; j++) {
            if (instances[j] >= sw.n_instances) { s->err = "instance out of range"; return MI355SAT_ERR_ARG; }
            if (!inst_open(sw, (uint32_t)instances[j])) continue;
            sw.dropped[instances[j]] = 1;
            sw.decided++;
            any = true;
        }
        if (any && sw.active && !sw.split && sw.decided < sw.n_instances) {
            if (sw.sts.size() != s->n_workers) gather_states(*s, sw.sts);
            rebalance_workers(*s, sw);
        }
        return 0;
    } catch (HipErr& he) { s->err = he.msg; return MI355SAT_ERR_HIP; }
    catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
}

int mi355sat_sweep_set_weights(mi355sat* s, const double* weights, uint64_t n) {
    if (!s || !s->sweep || !weights) return MI355SAT_ERR_STATE;
    Sweep& sw = s->sweep->sw;
    if (n != sw.n_instances) { s->err = "one weight per instance"; return MI355SAT_ERR_ARG; }
    std::vector<double> w(weights, weights + n);
    for (double x : w) if (!(x >= 0)) { s->err = "weights must be >= 0"; return MI355SAT_ERR_ARG; }
    if (w != sw.weights) { sw.weights.swap(w); sw.weights_dirty = true; }
    return 0;
}

int mi355sat_sweep_reopen(mi355sat* s, const uint64_t* instances, uint64_t n) {
    if (!s || !s->sweep || (n && !instances)) return MI355SAT_ERR_STATE;
    try {
        HIPCHK(hipSetDevice(s->device));
        Sweep& sw = s->sweep->sw;
        bool any = false;
        for (uint64_t j = 0; j < n; j++) {
            if (instances[j] >= sw.n_instances) { s->err = "instance out of range"; return MI355SAT_ERR_ARG; }
            const uint64_t i = instances[j];
            if (!sw.dropped[i] || sw.results[i] != MI355SAT_INTERRUPTED) continue;   // only what was withdrawn undecided
            sw.dropped[i] = 0;
            sw.decided--;
            any = true;
        }
        if (any && sw.active && !sw.split) {
            if (sw.sts.size() != s->n_workers) gather_states(*s, sw.sts);
            rebalance_workers(*s, sw);   // parked workers and those of decided / withdrawn instances take them up
        }
        return 0;
    } catch (HipErr& he) { s->err = he.msg; return MI355SAT_ERR_HIP; }
    catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
}

int mi355sat_sweep_model_of(mi355sat* s, uint64_t instance, int8_t* out, uint64_t n_vars) {
    if (!s || !s->sweep || !out) return MI355SAT_ERR_STATE;
    try {
        HIPCHK(hipSetDevice(s->device));
        Sweep& sw = s->sweep->sw;
        if (instance >= sw.n_instances || sw.results[instance] != MI355SAT_SAT || sw.winner[instance] < 0) {
            s->err = "no model for that instance";
            return MI355SAT_ERR_STATE;
        }
        std::vector<int8_t> m;
        fetch_model(*s, (uint32_t)sw.winner[instance], m, s->max_var);
        for (uint64_t v = 0; v < n_vars; v++) out[v] = v < m.size() ? m[v] : 0;
        return 0;
    } catch (HipErr& he) { s->err = he.msg; return MI355SAT_ERR_HIP; }
    catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
}

int mi355sat_sweep_end(mi355sat* s) {
    if (!s || !s->sweep) return MI355SAT_ERR_STATE;
    try {
        Sweep& sw = s->sweep->sw;
        s->batch_models.assign(sw.n_instances, {});
        for (uint32_t i = 0; i < sw.n_instances; i++)
            if (sw.results[i] == MI355SAT_SAT && sw.winner[i] >= 0)
                fetch_model(*s, (uint32_t)sw.winner[i], s->batch_models[i], s->max_var);
        consume_interrupt(*s);
        delete s->sweep;
        s->sweep = nullptr;
        return 0;
    } catch (HipErr& he) { s->err = he.msg; return MI355SAT_ERR_HIP; }
}

int32_t mi355sat_val(mi355sat* s, int32_t lit) {
    if (!s || lit == 0) return 0;
    uint64_t v = (uint64_t)(lit < 0 ? -(int64_t)lit : lit);
    if (v > s->model.size()) return 0;
    int8_t m = s->model[v - 1];
    if (m == 0) return 0;
    bool is_true = (lit > 0) == (m > 0);
    return is_true ? lit : -lit;
}

int mi355sat_model(mi355sat* s, int8_t* out, uint64_t n_vars) {
    if (!s || !out) return MI355SAT_ERR_ARG;
    if (s->model.empty() && s->max_var) { s->err = "no model (last result was not SAT)"; return MI355SAT_ERR_STATE; }
    for (uint64_t v = 0; v < n_vars; v++) out[v] = v < s->model.size() ? s->model[v] : 0;
    return 0;
}

int mi355sat_model_of(mi355sat* s, uint64_t instance, int8_t* out, uint64_t n_vars) {
    if (!s || !out) return MI355SAT_ERR_ARG;
    if (instance >= s->batch_models.size() || s->batch_models[instance].empty()) {
        s->err = "no model for that instance";
        return MI355SAT_ERR_STATE;
    }
    const auto& m = s->batch_models[instance];
    for (uint64_t v = 0; v < n_vars; v++) out[v] = v < m.size() ? m[v] : 0;
    return 0;
}

int mi355sat_debug_share_ring(mi355sat* s, int32_t* out, uint64_t cap_words, uint64_t* n_records) {
    if (!s || !n_records) return MI355SAT_ERR_ARG;
    *n_records = 0;
    if (!s->share_slots || !s->d_share_pool.p) return 0;
    try {
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipStreamSynchronize(s->stream));
        unsigned long long n = 0;
        HIPCHK(hipMemcpy(&n, s->d_share_n.p, sizeof n, hipMemcpyDeviceToHost));
        const uint64_t live = std::min<uint64_t>(n, s->share_slots);
        std::vector<int32_t> ring((size_t)live * MS_SHARE_REC);
        if (live) HIPCHK(hipMemcpy(ring.data(), s->d_share_pool.p, ring.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<uint32_t> inv(s->perm.size());
        for (uint32_t e = 0; e < s->perm.size(); e++) inv[s->perm[e]] = e;
        uint64_t w = 0;
        for (uint64_t r = 0; r < live; r++) {
            const int32_t* rec = ring.data() + r * MS_SHARE_REC;
            const int sz = rec[0] & 63;
            if (sz < 1 || sz > MS_SHARE_MAXLEN) { s->err = "malformed record in the exchange ring"; return MI355SAT_ERR_STATE; }
            if (out && w + (uint64_t)sz + 1 <= cap_words) {
                for (int j = 1; j <= sz; j++) {
                    if (rec[j] < 0 || (uint32_t)(rec[j] >> 1) >= inv.size()) { s->err = "literal out of range in the exchange ring"; return MI355SAT_ERR_STATE; }
                    out[w + j - 1] = (rec[j] & 1) ? -((int32_t)inv[rec[j] >> 1] + 1) : ((int32_t)inv[rec[j] >> 1] + 1);
                }
                out[w + sz] = 0;
            }
            w += (uint64_t)sz + 1;
            (*n_records)++;
        }
        return w <= cap_words || !out ? 0 : MI355SAT_ERR_ARG;
    } catch (HipErr& he) { s->err = he.msg; return MI355SAT_ERR_HIP; }
    catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
}

// ---- clause exchange between handles (GPUs) working on the SAME formula ------------------------------------------------
// Records travel as [lbd, DIMACS literals ..., 0] in the CALLER's variables: a ring record is a consequence of the
// simplified formula this handle searches, hence of the caller's formula (simplification only derives consequences,
// substituted variables are written as their representatives), so another handle - whose own simplification may have
// come out differently - can map it into its own variables and attach it.
#define MS_FOREIGN_PRODUCER 0x3ffffu    // producer id of records that came from another handle: never handed out again
int mi355sat_share_export(mi355sat* s, int32_t* out, uint64_t cap_words, uint64_t* n_words, uint64_t* n_records) {
    if (!s || !n_words || !n_records) return MI355SAT_ERR_ARG;
    *n_words = 0; *n_records = 0;
    if (!s->share_slots || !s->d_share_pool.p || !s->sweep) return 0;
    try {
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipStreamSynchronize(s->stream));
        unsigned long long n = 0;
        HIPCHK(hipMemcpy(&n, s->d_share_n.p, sizeof n, hipMemcpyDeviceToHost));
        uint64_t from = s->share_export_pos;
        if (n - from > s->share_slots) from = n - s->share_slots;     // overwritten before anybody asked
        if (n == from) return 0;
        std::vector<int32_t> recs((size_t)(n - from) * MS_SHARE_REC);
        for (uint64_t r = from; r < n;) {      // at most two pieces (the ring wraps)
            const uint64_t slot = r % s->share_slots, cnt = std::min<uint64_t>(n - r, s->share_slots - slot);
            HIPCHK(hipMemcpy(recs.data() + (r - from) * MS_SHARE_REC, s->d_share_pool.p + slot * MS_SHARE_REC,
                             cnt * MS_SHARE_REC * sizeof(int32_t), hipMemcpyDeviceToHost));
            r += cnt;
        }
        std::vector<uint32_t> inv(s->perm.size());
        for (uint32_t e = 0; e < s->perm.size(); e++) inv[s->perm[e]] = e;
        uint64_t w = 0, consumed = from;
        for (uint64_t r = from; r < n; r++) {
            const int32_t* rec = recs.data() + (r - from) * MS_SHARE_REC;
            const uint32_t hdr = (uint32_t)rec[0];
            const int sz = (int)(hdr & 63u);
            if ((hdr >> 14) == MS_FOREIGN_PRODUCER || sz < 1 || sz > MS_SHARE_MAXLEN) { consumed = r + 1; continue; }
            if (out && w + (uint64_t)sz + 2 > cap_words) break;        // the rest waits for the next call
            if (out) {
                out[w] = (int32_t)std::max<uint32_t>(1u, (hdr >> 6) & 255u);
                for (int j = 1; j <= sz; j++) {
                    if (rec[j] < 0 || (uint32_t)(rec[j] >> 1) >= inv.size()) { s->err = "literal out of range in the exchange ring"; return MI355SAT_ERR_STATE; }
                    out[w + j] = (rec[j] & 1) ? -((int32_t)inv[rec[j] >> 1] + 1) : ((int32_t)inv[rec[j] >> 1] + 1);
                }
                out[w + sz + 1] = 0;
            }
            w += (uint64_t)sz + 2;
            (*n_records)++;
            consumed = r + 1;
        }
        if (out) s->share_export_pos = consumed;     // (out == NULL only sizes the buffer)
        *n_words = w;
        return 0;
    } catch (HipErr& he) { s->err = he.msg; return MI355SAT_ERR_HIP; }
    catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
}

int mi355sat_share_import(mi355sat* s, const int32_t* clauses, uint64_t n_words, uint64_t* n_records) {
    if (!s || (!clauses && n_words)) return MI355SAT_ERR_ARG;
    if (n_records) *n_records = 0;
    if (!n_words || !s->share_slots || !s->d_share_pool.p || !s->sweep) return 0;
    try {
        HIPCHK(hipSetDevice(s->device));
        std::vector<uint8_t> gone(s->n_vars, 0);       // variables this handle's simplification resolved away
        for (const MsElim& e : s->elims) if ((uint32_t)(e.x >> 1) < gone.size()) gone[e.x >> 1] = 1;
        std::vector<int32_t> recs;
        uint64_t n_new = 0;
        for (uint64_t i = 0; i < n_words;) {
            const int32_t lbd = clauses[i++];
            int32_t rec[MS_SHARE_REC];
            int sz = 0;
            bool ok = lbd >= 1;
            while (i < n_words && clauses[i] != 0) {
                const int32_t d = clauses[i++];
                const uint64_t v = (uint64_t)(d < 0 ? -(int64_t)d : d);
                if (v == 0 || v > s->n_vars || sz >= MS_SHARE_MAXLEN) { ok = false; continue; }
                int32_t l = to_internal(d);
                while ((size_t)(l >> 1) < s->subst.size() && s->subst[l >> 1] != 2 * (l >> 1)) l = s->subst[l >> 1] ^ (l & 1);
                if (gone[l >> 1]) { ok = false; continue; }
                rec[1 + sz++] = 2 * (int32_t)s->perm[l >> 1] | (l & 1);
            }
            if (i >= n_words) { s->err = "share_import: clause without terminator"; return MI355SAT_ERR_ARG; }
            i++;   // the terminator
            if (!ok || sz < 1) continue;
            rec[0] = (int32_t)((uint32_t)sz | ((uint32_t)std::min<int32_t>(lbd, 255) << 6) | (MS_FOREIGN_PRODUCER << 14));
            for (int j = sz + 1; j < MS_SHARE_REC; j++) rec[j] = 0;
            recs.insert(recs.end(), rec, rec + MS_SHARE_REC);
            n_new++;
        }
        if (!n_new) return 0;
        if (n_new > s->share_slots) { s->err = "share_import: more records than the ring holds"; return MI355SAT_ERR_ARG; }
        HIPCHK(hipStreamSynchronize(s->stream));
        unsigned long long n = 0;
        HIPCHK(hipMemcpy(&n, s->d_share_n.p, sizeof n, hipMemcpyDeviceToHost));
        for (uint64_t r = 0; r < n_new;) {
            const uint64_t slot = (n + r) % s->share_slots, cnt = std::min<uint64_t>(n_new - r, s->share_slots - slot);
            HIPCHK(hipMemcpy(s->d_share_pool.p + slot * MS_SHARE_REC, recs.data() + r * MS_SHARE_REC,
                             cnt * MS_SHARE_REC * sizeof(int32_t), hipMemcpyHostToDevice));
            r += cnt;
        }
        n += n_new;
        HIPCHK(hipMemcpy(s->d_share_n.p, &n, sizeof n, hipMemcpyHostToDevice));
        if (n_records) *n_records = n_new;
        return 0;
    } catch (HipErr& he) { s->err = he.msg; return MI355SAT_ERR_HIP; }
    catch (std::bad_alloc&) { s->err = "out of host memory"; return MI355SAT_ERR_OOM; }
}

int mi355sat_stats(const mi355sat* s, mi355sat_stats_t* out) {
    if (!s || !out) return MI355SAT_ERR_ARG;
    *out = s->stats;
    return 0;
}

}  // extern "C"
