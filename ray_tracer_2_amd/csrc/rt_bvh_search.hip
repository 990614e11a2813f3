// rt_bvh_search.hip -- the SAH search of the reference's BVH builder on the GPU.
//
// `BVH::build` (src/core/bvh.rs:208-470) spends its time in `find_best_split` (:299-351):
// up to 3 x 50 candidate planes per node, each priced by `evaluate_sah` (:352-370), a pass
// over the node's triangles -- O(150 n) per tree level, ~13 s for the 1.05 M-triangle mesh of
// config 5 on one host core.  Everything else (the in-place partition :385-400 that fixes the
// triangle order, hence the traversal order, hence the floats the shader produces) is O(n)
// per level and stays on the host, operation for operation (csrc/host/bvh.cpp).
//
// evaluate_sah is order-independent bit for bit: the two sides' bounds are minima/maxima
// (exact in any order), the counts are integers below 2^24 (exact as f32 in any order), and
// the cost `nL * halfArea(L) + nR * halfArea(R)` is evaluated once from them with the
// reference's operations.  So a level's searches can be done here, all nodes and all
// candidates at once, and the host applies `cost < best_cost` in the reference's candidate
// order.  One workgroup prices all 150 candidates of one 512-triangle chunk of one node:
// the chunk is staged in LDS and every candidate (= thread) scans it with its 13 running
// values in registers -- no atomics.  Nodes of one chunk are finished in the workgroup;
// larger nodes leave per-chunk partials that a second kernel combines.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "host/bvh.h"

namespace rt2 {
namespace {

constexpr uint32_t CHUNK = 512;        // triangles per workgroup
constexpr uint32_t THREADS = 192;      // >= 150 candidates
constexpr uint32_t CANDIDATES = 150;   // 3 axes x at most 50 planes (bvh.rs:320-323)
constexpr uint32_t PARTIAL_FLOATS = 13;

struct DQuery {  // one node of the level
    float aabb_min[3], aabb_max[3];
    uint32_t start, count;
};
struct DBlock {  // one workgroup's job
    uint32_t query;        // node
    uint32_t first, count; // chunk: positions [first, first + count) of the current triangle order
    uint32_t partial;      // 0xffffffff: the node fits this chunk (finish here); else index of this chunk's partials
};
struct DMulti {  // a node spread over several chunks
    uint32_t query, first_partial, n_partials, _pad;
};
struct DResult {
    int32_t axis;
    float pos, cost;
    uint32_t _pad;
};

// candidate t of a node: axis, position, validity -- bvh.rs:311-335, the same f32 operations
struct Candidate {
    int axis;
    float pos;
    bool valid;
};
__device__ Candidate candidate_of(const DQuery& q, uint32_t t) {
    Candidate c;
    c.axis = (int)(t / 50u);
    const uint32_t i = t % 50u;
    const float b0 = q.aabb_max[0] - q.aabb_min[0], b1 = q.aabb_max[1] - q.aabb_min[1], b2 = q.aabb_max[2] - q.aabb_min[2];
    const float max_axis = fmaxf(b0, fmaxf(b1, b2));
    const float axis_size = c.axis == 0 ? b0 : (c.axis == 1 ? b1 : b2);
    const float axis_min = q.aabb_min[c.axis];
    c.valid = false;
    c.pos = 0.0f;
    if (t >= CANDIDATES || axis_size == 0.0f) return c;
    const float cc = ceilf(axis_size / max_axis * 50.0f);
    uint32_t n_tests;
    if (!(cc == cc) || cc <= 0.0f) n_tests = 0;       // `as u32`: NaN -> 0, saturating
    else if (cc >= 4294967296.0f) n_tests = 0xffffffffu;
    else n_tests = (uint32_t)cc;
    if (n_tests < 1) n_tests = 1;
    if (n_tests > 50) n_tests = 50;
    if (i >= n_tests) return c;
    const float split_t = (float)(i + 1u) / ((float)n_tests + 1.0f);
    c.pos = axis_min + axis_size * split_t;
    c.valid = true;
    return c;
}

struct Sides {  // evaluate_sah's running values
    float nl, nr;
    float lmn[3], lmx[3], rmn[3], rmx[3];
};
__device__ void sides_init(Sides& s) {
    s.nl = s.nr = 0.0f;
    for (int k = 0; k < 3; ++k) {
        s.lmn[k] = s.rmn[k] = INFINITY;
        s.lmx[k] = s.rmx[k] = -INFINITY;
    }
}
__device__ float sides_cost(const Sides& s) {  // bvh.rs:366-369 (+ Aabb::half_area :87-90)
    const float lex = s.lmx[0] - s.lmn[0], ley = s.lmx[1] - s.lmn[1], lez = s.lmx[2] - s.lmn[2];
    const float rex = s.rmx[0] - s.rmn[0], rey = s.rmx[1] - s.rmn[1], rez = s.rmx[2] - s.rmn[2];
    const float lha = (lex * ley + ley * lez) + lex * lez;
    const float rha = (rex * rey + rey * rez) + rex * rez;
    return s.nl * lha + s.nr * rha;
}

// the reference's `if cost < best_cost` over the candidates in its order (axis outer, plane inner)
__device__ void pick_best(const float* cost, const float* pos, const uint32_t* valid, DResult& out) {
    float best = INFINITY;
    int axis = 0;
    float split = 0.0f;
    for (uint32_t t = 0; t < CANDIDATES; ++t) {
        if (valid[t] && cost[t] < best) {
            best = cost[t];
            axis = (int)(t / 50u);
            split = pos[t];
        }
    }
    out.axis = axis;
    out.pos = split;
    out.cost = best;
    out._pad = 0;
}

__global__ void __launch_bounds__(THREADS) sah_chunks_kernel(const float* __restrict__ tri9,  // per triangle id: centroid, min, max
                                                             const uint32_t* __restrict__ order, const DQuery* __restrict__ queries,
                                                             const DBlock* __restrict__ blocks, float* __restrict__ partials,
                                                             DResult* __restrict__ results) {
    __shared__ float lds_tri[CHUNK * 9];
    __shared__ float lds_cost[CANDIDATES], lds_pos[CANDIDATES];
    __shared__ uint32_t lds_valid[CANDIDATES];
    const DBlock job = blocks[blockIdx.x];
    const DQuery q = queries[job.query];
    for (uint32_t k = threadIdx.x; k < job.count * 9u; k += THREADS) {
        const uint32_t tri = order[job.first + k / 9u];
        lds_tri[k] = tri9[(size_t)tri * 9u + k % 9u];
    }
    __syncthreads();
    const uint32_t t = threadIdx.x;
    const Candidate c = candidate_of(q, t);
    Sides s;
    sides_init(s);
    if (c.valid) {
        for (uint32_t k = 0; k < job.count; ++k) {
            const float* tr = lds_tri + k * 9u;
            if (tr[c.axis] < c.pos) {  // bvh.rs:359
                s.nl += 1.0f;
                for (int d = 0; d < 3; ++d) {
                    s.lmn[d] = fminf(s.lmn[d], tr[3 + d]);
                    s.lmx[d] = fmaxf(s.lmx[d], tr[6 + d]);
                }
            } else {
                s.nr += 1.0f;
                for (int d = 0; d < 3; ++d) {
                    s.rmn[d] = fminf(s.rmn[d], tr[3 + d]);
                    s.rmx[d] = fmaxf(s.rmx[d], tr[6 + d]);
                }
            }
        }
    }
    if (job.partial != 0xffffffffu) {
        if (t < CANDIDATES) {
            float* p = partials + ((size_t)job.partial * CANDIDATES + t) * PARTIAL_FLOATS;
            p[0] = s.nl;
            for (int d = 0; d < 3; ++d) {
                p[1 + d] = s.lmn[d];
                p[4 + d] = s.lmx[d];
                p[7 + d] = s.rmn[d];
                p[10 + d] = s.rmx[d];
            }
        }
        return;
    }
    if (t < CANDIDATES) {
        lds_cost[t] = c.valid ? sides_cost(s) : INFINITY;
        lds_pos[t] = c.pos;
        lds_valid[t] = c.valid ? 1u : 0u;
    }
    __syncthreads();
    if (t == 0) pick_best(lds_cost, lds_pos, lds_valid, results[job.query]);
}

__global__ void __launch_bounds__(THREADS) sah_combine_kernel(const DQuery* __restrict__ queries, const DMulti* __restrict__ multis,
                                                              const float* __restrict__ partials, DResult* __restrict__ results) {
    __shared__ float lds_cost[CANDIDATES], lds_pos[CANDIDATES];
    __shared__ uint32_t lds_valid[CANDIDATES];
    const DMulti m = multis[blockIdx.x];
    const DQuery q = queries[m.query];
    const uint32_t t = threadIdx.x;
    const Candidate c = candidate_of(q, t);
    if (t < CANDIDATES) {
        Sides s;
        sides_init(s);
        if (c.valid) {
            for (uint32_t b = 0; b < m.n_partials; ++b) {
                const float* p = partials + ((size_t)(m.first_partial + b) * CANDIDATES + t) * PARTIAL_FLOATS;
                s.nl += p[0];  // integers below 2^24: exact
                for (int d = 0; d < 3; ++d) {
                    s.lmn[d] = fminf(s.lmn[d], p[1 + d]);
                    s.lmx[d] = fmaxf(s.lmx[d], p[4 + d]);
                    s.rmn[d] = fminf(s.rmn[d], p[7 + d]);
                    s.rmx[d] = fmaxf(s.rmx[d], p[10 + d]);
                }
            }
            s.nr = (float)q.count - s.nl;  // exact
        }
        lds_cost[t] = c.valid ? sides_cost(s) : INFINITY;
        lds_pos[t] = c.pos;
        lds_valid[t] = c.valid ? 1u : 0u;
    }
    __syncthreads();
    if (t == 0) pick_best(lds_cost, lds_pos, lds_valid, results[m.query]);
}

#define HIP_OK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e_)); \
    } while (0)

struct DeviceSearch {
    float* tri9 = nullptr;
    uint32_t* order = nullptr;
    DQuery* queries = nullptr;
    DBlock* blocks = nullptr;
    DMulti* multis = nullptr;
    float* partials = nullptr;
    DResult* results = nullptr;
    size_t cap_q = 0, cap_r = 0, cap_b = 0, cap_m = 0, cap_p = 0, n = 0;
    hipStream_t stream = nullptr;

    ~DeviceSearch() {
        for (void* p : {(void*)tri9, (void*)order, (void*)queries, (void*)blocks, (void*)multis, (void*)partials, (void*)results})
            if (p) (void)hipFree(p);
        if (stream) (void)hipStreamDestroy(stream);
    }
    template <class T>
    void grow(T*& p, size_t& cap, size_t need) {
        if (need <= cap) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = need + need / 2 + 16;
        HIP_OK(hipMalloc((void**)&p, cap * sizeof(T)));
    }
};

}  // namespace

LevelSearch make_device_level_search(int device, const float* tri9, size_t n_tris) {
    auto ds = std::make_shared<DeviceSearch>();
    HIP_OK(hipSetDevice(device));
    HIP_OK(hipStreamCreate(&ds->stream));
    ds->n = n_tris;
    HIP_OK(hipMalloc((void**)&ds->tri9, n_tris * 9 * sizeof(float)));
    HIP_OK(hipMalloc((void**)&ds->order, n_tris * sizeof(uint32_t)));
    HIP_OK(hipMemcpyAsync(ds->tri9, tri9, n_tris * 9 * sizeof(float), hipMemcpyHostToDevice, ds->stream));
    return [ds, device](const uint32_t* order, size_t n, const std::vector<SplitQuery>& qs, std::vector<SplitResult>& out) {
        out.resize(qs.size());
        if (qs.empty()) return;
        HIP_OK(hipSetDevice(device));
        std::vector<DQuery> dq(qs.size());
        std::vector<DBlock> db;
        std::vector<DMulti> dm;
        uint32_t n_partials = 0;
        for (size_t k = 0; k < qs.size(); ++k) {
            memcpy(dq[k].aabb_min, qs[k].aabb_min, 12);
            memcpy(dq[k].aabb_max, qs[k].aabb_max, 12);
            dq[k].start = qs[k].start;
            dq[k].count = qs[k].count;
            const uint32_t chunks = (qs[k].count + CHUNK - 1) / CHUNK;
            if (chunks > 1) dm.push_back(DMulti{(uint32_t)k, n_partials, chunks, 0});
            for (uint32_t c = 0; c < chunks; ++c) {
                const uint32_t first = qs[k].start + c * CHUNK;
                const uint32_t cnt = std::min(CHUNK, qs[k].start + qs[k].count - first);
                db.push_back(DBlock{(uint32_t)k, first, cnt, chunks > 1 ? n_partials++ : 0xffffffffu});
            }
        }
        ds->grow(ds->queries, ds->cap_q, dq.size());
        ds->grow(ds->results, ds->cap_r, dq.size());
        ds->grow(ds->blocks, ds->cap_b, db.size());
        ds->grow(ds->multis, ds->cap_m, dm.size());
        ds->grow(ds->partials, ds->cap_p, (size_t)n_partials * CANDIDATES * PARTIAL_FLOATS);
        HIP_OK(hipMemcpyAsync(ds->order, order, n * sizeof(uint32_t), hipMemcpyHostToDevice, ds->stream));
        HIP_OK(hipMemcpyAsync(ds->queries, dq.data(), dq.size() * sizeof(DQuery), hipMemcpyHostToDevice, ds->stream));
        HIP_OK(hipMemcpyAsync(ds->blocks, db.data(), db.size() * sizeof(DBlock), hipMemcpyHostToDevice, ds->stream));
        if (!dm.empty()) HIP_OK(hipMemcpyAsync(ds->multis, dm.data(), dm.size() * sizeof(DMulti), hipMemcpyHostToDevice, ds->stream));
        hipLaunchKernelGGL(sah_chunks_kernel, dim3((uint32_t)db.size()), dim3(THREADS), 0, ds->stream, ds->tri9, ds->order, ds->queries,
                           ds->blocks, ds->partials, ds->results);
        if (!dm.empty())
            hipLaunchKernelGGL(sah_combine_kernel, dim3((uint32_t)dm.size()), dim3(THREADS), 0, ds->stream, ds->queries, ds->multis,
                               ds->partials, ds->results);
        HIP_OK(hipGetLastError());
        std::vector<DResult> dr(dq.size());
        HIP_OK(hipMemcpyAsync(dr.data(), ds->results, dr.size() * sizeof(DResult), hipMemcpyDeviceToHost, ds->stream));
        HIP_OK(hipStreamSynchronize(ds->stream));
        for (size_t k = 0; k < dr.size(); ++k) {
            out[k].axis = dr[k].axis;
            out[k].pos = dr[k].pos;
            out[k].cost = dr[k].cost;
        }
    };
}

}  // namespace rt2
