// Native replay list: re-issue the kernels of a stream-captured step from a C loop.
//
// The launch-bound regimes of the path (config C1: ~700 launch-latency-sized kernels per timestep at batch 4; config C3's 4
// images per GPU; the 40 UNet forwards of an LDM importance step) pay 30-90 us of Python / ctypes per launch, and on this stack
// hipGraphLaunch of the same ~750-node graph costs MORE host time than the eager launches.  A HIP
// stream capture is still the right RECORDER -- it sees every kernel, memset and cross-stream edge of the step, whoever
// launched it, and torch's capture pool pins the memory the step touches -- so the step is captured once into a hipGraph,
// never instantiated, and this file walks the graph: node parameters are read back with hipGraph*NodeGetParams, the nodes are
// list-scheduled onto two streams along the captured dependency edges (cross-stream edges become event record / wait pairs),
// and dp_replay_launch re-issues them with hipLaunchKernel -- a few microseconds per node, same kernels, same arguments,
// same order per dependency chain, hence the same bits as the eager step.
//
// The graph object must outlive the replay list (kernel argument storage belongs to the graph).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>
#include "../../include/dp_hip.h"

namespace {

enum NodeKind { KERNEL = 0, MEMSET = 1, MEMCPY = 2, NOP = 3 };

struct RNode {
    NodeKind kind;
    int stream;                          // 0 / 1
    int record;                          // index into events to record after this node, or -1
    std::vector<int> waits;              // events to wait for before this node
    hipKernelNodeParams k;
    hipMemsetParams ms;
    hipMemcpy3DParms cp;
};

struct Replay {
    std::vector<RNode> nodes;
    std::vector<hipEvent_t> events;
    hipEvent_t fork = nullptr, join = nullptr;
    int n_kernel = 0, n_memset = 0, n_memcpy = 0, n_nop = 0, n_cross = 0, n_on_side = 0;
};

#define RP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = (int)e_; goto fail; } } while (0)

}  // namespace

extern "C" int dp_replay_build(void* graph_v, void** out) {
    hipGraph_t graph = (hipGraph_t)graph_v;
    int err = 0;
    Replay* rp = new Replay();
    size_t n = 0;
    std::vector<hipGraphNode_t> gn;
    std::unordered_map<hipGraphNode_t, int> index;
    std::vector<std::vector<int>> deps;
    std::vector<int> order, stream_of, pos_in_order;
    {
        RP_CHECK(hipGraphGetNodes(graph, nullptr, &n));
        gn.resize(n);
        RP_CHECK(hipGraphGetNodes(graph, gn.data(), &n));
        for (size_t i = 0; i < n; ++i) index[gn[i]] = (int)i;
        deps.resize(n);
        std::vector<int> indeg(n, 0);
        std::vector<std::vector<int>> succ(n);
        for (size_t i = 0; i < n; ++i) {
            size_t nd = 0;
            RP_CHECK(hipGraphNodeGetDependencies(gn[i], nullptr, &nd));
            std::vector<hipGraphNode_t> d(nd);
            if (nd) RP_CHECK(hipGraphNodeGetDependencies(gn[i], d.data(), &nd));
            for (size_t j = 0; j < nd; ++j) {
                auto it = index.find(d[j]);
                if (it == index.end()) { err = (int)hipErrorInvalidValue; goto fail; }
                deps[i].push_back(it->second);
                succ[it->second].push_back((int)i);
            }
            indeg[i] = (int)nd;
        }
        // topological order that keeps the capture (creation) order wherever the edges allow: smallest ready index first
        std::vector<char> done(n, 0);
        order.reserve(n);
        {
            // creation order is already topological for stream captures; verify, else fall back to Kahn's algorithm
            bool topo = true;
            for (size_t i = 0; i < n && topo; ++i)
                for (int d : deps[i]) if (d >= (int)i) { topo = false; break; }
            if (topo) {
                for (size_t i = 0; i < n; ++i) order.push_back((int)i);
            } else {
                std::vector<int> ready;
                for (size_t i = 0; i < n; ++i) if (!indeg[i]) ready.push_back((int)i);
                while (!ready.empty()) {
                    size_t best = 0;
                    for (size_t j = 1; j < ready.size(); ++j) if (ready[j] < ready[best]) best = j;
                    const int v = ready[best];
                    ready.erase(ready.begin() + best);
                    order.push_back(v);
                    for (int s : succ[v]) if (--indeg[s] == 0) ready.push_back(s);
                }
                if (order.size() != n) { err = (int)hipErrorInvalidValue; goto fail; }
            }
        }
        // list scheduling onto two streams: a node continues the stream whose tail it depends on; a node none of whose
        // dependencies is a tail starts on the OTHER stream than its first dependency (that is the captured fork: the first
        // kernel of the forked work continued the tail, the original chain goes on beside it)
        stream_of.assign(n, 0);
        pos_in_order.assign(n, 0);
        const bool one_stream = getenv("DP_REPLAY_ONE_STREAM") != nullptr;     // experiment: capture order on one stream
        int tail[2] = {-1, -1};
        std::vector<int> ev_of(n, -1);
        int synced[2][2] = {{-1, -1}, {-1, -1}};      // synced[s][t]: stream s has waited for stream t's nodes up to this order position
        rp->nodes.resize(n);
        for (size_t oi = 0; oi < n; ++oi) {
            const int i = order[oi];
            pos_in_order[i] = (int)oi;
            int s = -1;
            for (int d : deps[i]) {
                const int ds = stream_of[d];
                if (tail[ds] == d && (s < 0 || ds < s)) s = ds;
            }
            if (s < 0) s = deps[i].empty() ? 0 : (stream_of[deps[i][0]] ^ 1);
            if (one_stream) s = 0;
            stream_of[i] = s;
            RNode& r = rp->nodes[oi];
            r.stream = s;
            r.record = -1;
            for (int d : deps[i]) {
                const int ds = stream_of[d];
                if (ds == s) continue;                               // stream order
                if (synced[s][ds] >= pos_in_order[d]) continue;      // an earlier node of this stream already waited past it
                if (ev_of[d] < 0) {
                    ev_of[d] = (int)rp->events.size();
                    rp->events.push_back(nullptr);
                    rp->nodes[pos_in_order[d]].record = ev_of[d];
                }
                r.waits.push_back(ev_of[d]);
                synced[s][ds] = pos_in_order[d];
                ++rp->n_cross;
            }
            tail[s] = i;
            hipGraphNodeType ty;
            RP_CHECK(hipGraphNodeGetType(gn[i], &ty));
            if (ty == hipGraphNodeTypeKernel) {
                r.kind = KERNEL;
                memset(&r.k, 0, sizeof(r.k));
                RP_CHECK(hipGraphKernelNodeGetParams(gn[i], &r.k));
                if (!r.k.func || !r.k.kernelParams) { err = (int)hipErrorNotSupported; goto fail; }     // `extra`-style launches: not re-issued
                {   // a node captured from hipModuleLaunchKernel carries a hipFunction_t, which hipLaunchKernel cannot launch: such
                    // a failure would surface part-way through a re-issued timestep (gradients half accumulated).  Only host-side
                    // kernel symbols resolve here, so refuse the list now.
                    hipFuncAttributes fa;
                    if (hipFuncGetAttributes(&fa, r.k.func) != hipSuccess) {
                        (void)hipGetLastError();
                        err = (int)hipErrorNotSupported;
                        goto fail;
                    }
                }
                ++rp->n_kernel;
            } else if (ty == hipGraphNodeTypeMemset) {
                r.kind = MEMSET;
                RP_CHECK(hipGraphMemsetNodeGetParams(gn[i], &r.ms));
                if (r.ms.height > 1 || (r.ms.elementSize != 1 && r.ms.elementSize != 2 && r.ms.elementSize != 4)) {
                    err = (int)hipErrorNotSupported;
                    goto fail;
                }
                ++rp->n_memset;
            } else if (ty == hipGraphNodeTypeMemcpy) {
                r.kind = MEMCPY;
                memset(&r.cp, 0, sizeof(r.cp));
                RP_CHECK(hipGraphMemcpyNodeGetParams(gn[i], &r.cp));
                ++rp->n_memcpy;
            } else if (ty == hipGraphNodeTypeEmpty) {
                r.kind = NOP;
                ++rp->n_nop;
            } else {
                err = (int)hipErrorNotSupported;                     // host / event / mem-alloc nodes: not a step we can re-issue
                goto fail;
            }
            if (s == 1) ++rp->n_on_side;
        }
        for (auto& e : rp->events) RP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        RP_CHECK(hipEventCreateWithFlags(&rp->fork, hipEventDisableTiming));
        RP_CHECK(hipEventCreateWithFlags(&rp->join, hipEventDisableTiming));
    }
    *out = rp;
    return 0;
fail:
    for (auto e : rp->events) if (e) (void)hipEventDestroy(e);
    if (rp->fork) (void)hipEventDestroy(rp->fork);
    if (rp->join) (void)hipEventDestroy(rp->join);
    delete rp;
    *out = nullptr;
    return err ? err : (int)hipErrorUnknown;
}

extern "C" int dp_replay_launch(void* handle, void* main_stream, void* side_stream) {
    Replay* rp = (Replay*)handle;
    if (!rp) return (int)hipErrorInvalidValue;
    hipStream_t st[2] = {(hipStream_t)main_stream, (hipStream_t)side_stream};
    const bool two = rp->n_on_side > 0;
    hipError_t e = hipSuccess;
    if (two) {                                                     // the side stream starts behind everything enqueued so far
        if ((e = hipEventRecord(rp->fork, st[0])) != hipSuccess) return (int)e;
        if ((e = hipStreamWaitEvent(st[1], rp->fork, 0)) != hipSuccess) return (int)e;
    }
    for (RNode& r : rp->nodes) {
        hipStream_t s = st[r.stream];
        for (int w : r.waits)
            if ((e = hipStreamWaitEvent(s, rp->events[w], 0)) != hipSuccess) return (int)e;
        switch (r.kind) {
        case KERNEL:
            if (r.k.kernelParams)
                e = hipLaunchKernel(r.k.func, r.k.gridDim, r.k.blockDim, r.k.kernelParams, r.k.sharedMemBytes, s);
            else
                e = hipErrorNotSupported;
            break;
        case MEMSET:
            if (r.ms.elementSize == 1)      e = hipMemsetAsync(r.ms.dst, (int)r.ms.value, r.ms.width, s);
            else if (r.ms.elementSize == 2) e = hipMemsetD16Async((hipDeviceptr_t)r.ms.dst, (unsigned short)r.ms.value, r.ms.width, s);
            else                            e = hipMemsetD32Async((hipDeviceptr_t)r.ms.dst, (int)r.ms.value, r.ms.width, s);
            break;
        case MEMCPY:
            e = hipMemcpy3DAsync(&r.cp, s);
            break;
        case NOP:
            break;
        }
        if (e != hipSuccess) return (int)e;
        if (r.record >= 0 && (e = hipEventRecord(rp->events[r.record], s)) != hipSuccess) return (int)e;
    }
    if (two) {                                                     // join: the caller's stream continues behind both
        if ((e = hipEventRecord(rp->join, st[1])) != hipSuccess) return (int)e;
        if ((e = hipStreamWaitEvent(st[0], rp->join, 0)) != hipSuccess) return (int)e;
    }
    return 0;
}

extern "C" int dp_replay_info(void* handle, int* out8) {
    Replay* rp = (Replay*)handle;
    if (!rp) return (int)hipErrorInvalidValue;
    out8[0] = (int)rp->nodes.size();
    out8[1] = rp->n_kernel;
    out8[2] = rp->n_memset;
    out8[3] = rp->n_memcpy;
    out8[4] = rp->n_nop;
    out8[5] = rp->n_cross;
    out8[6] = rp->n_on_side;
    out8[7] = (int)rp->events.size();
    return 0;
}

extern "C" int dp_replay_free(void* handle) {
    Replay* rp = (Replay*)handle;
    if (!rp) return 0;
    for (auto e : rp->events) if (e) (void)hipEventDestroy(e);
    if (rp->fork) (void)hipEventDestroy(rp->fork);
    if (rp->join) (void)hipEventDestroy(rp->join);
    delete rp;
    return 0;
}
