// Launch-list replay of a captured G+D iteration.
//
// The iteration (models/PDGNet_v2.py:171-256; pdgn_amd/trainer.py) is ~1300 kernel launches spread over seven HIP streams.
// Issued eagerly from Python the HOST needs 25-27 ms for them (autograd nodes, allocations, ctypes marshalling: ~20 us per
// launch), which is within 2-4 ms of what the DEVICE needs; hipGraphLaunch of the captured iteration is slower than the eager
// step on this ROCm (34.0 vs 31.2 ms: its executor re-derives streams and barriers on its own).  This file takes the third
// route: the iteration is CAPTURED once (stream capture, so every launch parameter and every cross-stream dependency is
// recorded by the runtime), the captured graph is read back node by node, and every later iteration RE-ISSUES the same
// launches with plain hipModuleLaunchKernel calls on the streams the capture used -- same kernels, same arguments, same
// streams and hardware queues as the eager step, at ~4 us of host time per launch and no Python in between.
//
// Stream identity is not part of a captured graph, so the capture tags its streams: pdgn_replay_marker(id, stream) launches an
// empty kernel whose argument is the stream's id.  Nodes are walked in capture order (a topological order: a captured node
// depends only on earlier ones); a node continues the chain of its first dependency that is still the tail of a chain, a
// marker node opens the chain of its id, anything else opens an anonymous chain (replayed on a spare stream).  Dependencies
// inside a chain are stream order; dependencies across chains become hipEventRecord / hipStreamWaitEvent pairs.
//
// Marker ids >= PDGN_REPLAY_POINT_BASE (64) do not name a stream: they are POINTS of the list at which the host does something the
// capture could not record -- under data parallelism the RCCL all-reduces of the gradient buffers (a collective cannot be
// captured).  Such a marker stays in the chain of the stream it was launched on; pdgn_replay_points reports its id, its position
// in the list and its chain, and the caller issues the list in ranges with its own calls in between (pdgn_amd/trainer.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/pdgn_hip.h"

namespace {

__global__ void replay_marker_kernel(int id) { (void)id; }

enum NodeKind { NK_KERNEL = 0, NK_MEMSET = 1, NK_MEMCPY = 2, NK_EMPTY = 3 };

struct RNode {
    int kind = NK_EMPTY;
    int chain = -1;
    int point = -1;                  // id of a host point (marker id >= 64), or -1
    void *sym = nullptr;             // kernel: the host symbol it was launched through (identifies a kernel instance)
    int tstart = -1, tstop = -1;     // timed span that begins in front of / ends behind this node, or -1
    int record = -1;                 // event to record after this node (another chain waits for it), or -1
    std::vector<int> waits;          // events this node's chain must wait for before the node
    // kernel
    hipFunction_t func = nullptr;
    unsigned gx = 1, gy = 1, gz = 1, bx = 1, by = 1, bz = 1, shmem = 0;
    void **params = nullptr;
    void **extra = nullptr;
    // memset / memcpy
    void *dst = nullptr;
    const void *src = nullptr;
    size_t bytes = 0;
    unsigned value = 0, elem = 1;
    size_t width = 0;
};

struct RPlan {
    std::vector<RNode> nodes;
    std::vector<int> chain_label;    // marker id of a chain, or -1
    std::vector<hipStream_t> chain_stream;
    std::vector<hipEvent_t> events;
    int counts[8] = {0};             // nodes, kernels, memsets, memcpys, empties, chains, events, labelled chains
    int joined = 0;                  // the last node of chain 'marker 0' has the last node of every other chain among its ancestors
    // in-iteration timing of chosen kernel nodes (bench.py's roofline: the dominant kernel's duration INSIDE the timed region)
    int time_slots = 0;
    std::vector<int> time_count;     // per timed span: passes recorded so far
    std::vector<hipEvent_t> time_ev; // [span][slot][start, stop]
    hipStream_t origin = nullptr;    // the capture's origin stream: chain of marker 0
};

bool is_marker(const hipKernelNodeParams &p) { return p.func == (void *)replay_marker_kernel; }

}  // namespace

extern "C" int pdgn_replay_marker(int id, pdgn_stream_t stream) {
    hipLaunchKernelGGL(replay_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, id);
    return (int)hipGetLastError();
}

extern "C" int pdgn_replay_build(void *graph_, void **plan_out) {
    hipGraph_t graph = (hipGraph_t)graph_;
    if (!graph || !plan_out) return PDGN_ERR_INVALID;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
    if (e != hipSuccess) return (int)e;
    std::vector<hipGraphNode_t> gn(n);
    if (n && (e = hipGraphGetNodes(graph, gn.data(), &n)) != hipSuccess) return (int)e;
    std::unordered_map<hipGraphNode_t, int> index;
    for (size_t i = 0; i < n; ++i) index[gn[i]] = (int)i;
    // dependencies per node, in the order the capture recorded them (own stream first, waited events after)
    std::vector<std::vector<int>> deps(n);
    for (size_t i = 0; i < n; ++i) {
        size_t nd = 0;
        if ((e = hipGraphNodeGetDependencies(gn[i], nullptr, &nd)) != hipSuccess) return (int)e;
        std::vector<hipGraphNode_t> d(nd);
        if (nd && (e = hipGraphNodeGetDependencies(gn[i], d.data(), &nd)) != hipSuccess) return (int)e;
        for (size_t j = 0; j < nd; ++j) {
            auto it = index.find(d[j]);
            if (it == index.end()) return PDGN_ERR_INVALID;
            deps[i].push_back(it->second);
        }
    }
    // a topological order that stays as close to capture order as the dependencies allow
    std::vector<int> order, indeg(n, 0), pos(n, -1);
    std::vector<std::vector<int>> succ(n);
    for (size_t i = 0; i < n; ++i)
        for (int d : deps[i]) { succ[d].push_back((int)i); ++indeg[i]; }
    {
        std::vector<int> heap;
        auto cmp = [](int a, int b) { return a > b; };
        for (size_t i = 0; i < n; ++i) if (!indeg[i]) heap.push_back((int)i);
        std::make_heap(heap.begin(), heap.end(), cmp);
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            int u = heap.back(); heap.pop_back();
            pos[u] = (int)order.size();
            order.push_back(u);
            for (int v : succ[u]) if (--indeg[v] == 0) { heap.push_back(v); std::push_heap(heap.begin(), heap.end(), cmp); }
        }
        if (order.size() != n) return PDGN_ERR_INVALID;       // a cycle: not a captured graph
    }
    RPlan *plan = new RPlan();
    plan->nodes.resize(n);
    std::vector<int> tail;                                    // tail[c] = node (original index) at the end of chain c
    std::vector<int> chain_of(n, -1);
    std::vector<int> record_of(n, -1);
    for (int u : order) {
        RNode &r = plan->nodes[pos[u]];
        hipGraphNodeType t;
        if ((e = hipGraphNodeGetType(gn[u], &t)) != hipSuccess) { delete plan; return (int)e; }
        int marker = -1;
        if (t == hipGraphNodeTypeKernel) {
            hipKernelNodeParams p;
            memset(&p, 0, sizeof(p));
            if ((e = hipGraphKernelNodeGetParams(gn[u], &p)) != hipSuccess) { delete plan; return (int)e; }
            r.kind = NK_KERNEL;
            r.gx = p.gridDim.x; r.gy = p.gridDim.y; r.gz = p.gridDim.z;
            r.bx = p.blockDim.x; r.by = p.blockDim.y; r.bz = p.blockDim.z;
            r.shmem = p.sharedMemBytes;
            r.params = p.kernelParams;
            r.extra = p.extra;
            r.sym = (void *)p.func;
            hipFunction_t f = nullptr;
            if (hipGetFuncBySymbol(&f, p.func) == hipSuccess && f) r.func = f;      // a __global__ function's host stub
            else { (void)hipGetLastError(); r.func = (hipFunction_t)p.func; }       // launched through the module API
            if (is_marker(p) && p.kernelParams) marker = *(int *)p.kernelParams[0];
            if (marker >= 64) { r.point = marker; marker = -1; }     // a host point: an ordinary node of its stream's chain
            ++plan->counts[1];
        } else if (t == hipGraphNodeTypeMemset) {
            hipMemsetParams p;
            memset(&p, 0, sizeof(p));
            if ((e = hipGraphMemsetNodeGetParams(gn[u], &p)) != hipSuccess) { delete plan; return (int)e; }
            if (p.height > 1) { delete plan; return PDGN_ERR_INVALID; }
            r.kind = NK_MEMSET;
            r.dst = p.dst; r.value = p.value; r.elem = p.elementSize; r.width = p.width;
            ++plan->counts[2];
        } else if (t == hipGraphNodeTypeMemcpy) {
            hipMemcpy3DParms p;
            memset(&p, 0, sizeof(p));
            if ((e = hipGraphMemcpyNodeGetParams(gn[u], &p)) != hipSuccess) { delete plan; return (int)e; }
            // only flat device-to-device copies are re-issued (what torch's copy_ of a contiguous tensor records)
            if (p.extent.height > 1 || p.extent.depth > 1 || !p.dstPtr.ptr || !p.srcPtr.ptr || p.srcArray || p.dstArray ||
                p.srcPos.x || p.srcPos.y || p.srcPos.z || p.dstPos.x || p.dstPos.y || p.dstPos.z ||
                !(p.kind == hipMemcpyDeviceToDevice || p.kind == hipMemcpyDefault)) { delete plan; return -2; }
            r.kind = NK_MEMCPY;
            r.dst = p.dstPtr.ptr; r.src = p.srcPtr.ptr; r.bytes = p.extent.width;
            ++plan->counts[3];
        } else if (t == hipGraphNodeTypeEmpty) {
            r.kind = NK_EMPTY;
            ++plan->counts[4];
        } else {
            delete plan;
            return -3 - (int)t;                                  // host / event / child-graph nodes: not part of this iteration
        }
        // chain assignment
        int c = -1;
        if (marker < 0)
            for (int d : deps[u]) { int cd = chain_of[d]; if (tail[cd] == d) { c = cd; break; } }
        if (c < 0) {
            c = (int)tail.size();
            tail.push_back(u);
            plan->chain_label.push_back(marker);
        }
        tail[c] = u;
        chain_of[u] = c;
        r.chain = c;
        for (int d : deps[u]) {
            if (chain_of[d] == c) continue;                       // stream order covers it (d is earlier in the same chain)
            if (record_of[d] < 0) {
                record_of[d] = (int)plan->events.size();
                plan->events.push_back(nullptr);
                plan->nodes[pos[d]].record = record_of[d];
            }
            if (std::find(r.waits.begin(), r.waits.end(), record_of[d]) == r.waits.end()) r.waits.push_back(record_of[d]);
        }
    }
    // Issue order.  Capture order issues whatever the Python schedule enqueued first -- e.g. the ~240 launches of the four
    // discriminators' real halves before the issuing stream's first kernel.  With PDGN_REPLAY_BURST = R > 0 the list is
    // re-ordered (still topologically): up to R nodes of the issuing stream's chain (marker 0) per node of any other chain
    // whenever both are ready, other chains in capture order among themselves.
    {
        const char *env = getenv("PDGN_REPLAY_BURST");
        const int burst = env ? atoi(env) : 0;
        int main_chain = -1;
        for (size_t c = 0; c < plan->chain_label.size(); ++c) if (plan->chain_label[c] == 0) main_chain = (int)c;
        if (burst > 0 && main_chain >= 0) {
            const int N = (int)n;
            // dependencies in list positions
            std::vector<std::vector<int>> pdeps(N), psucc(N);
            for (int u = 0; u < N; ++u)
                for (int d : deps[u]) { pdeps[pos[u]].push_back(pos[d]); psucc[pos[d]].push_back(pos[u]); }
            std::vector<int> left(N);
            for (int i = 0; i < N; ++i) left[i] = (int)pdeps[i].size();
            std::vector<char> done(N, 0);
            std::vector<int> neworder;
            neworder.reserve(N);
            int run = 0;
            size_t scan_main = 0, scan_side = 0;
            auto next_ready = [&](bool want_main, size_t &scan) {
                while (scan < (size_t)N && (done[scan] || (plan->nodes[scan].chain == main_chain) != want_main)) ++scan;
                for (size_t i = scan; i < (size_t)N; ++i)
                    if (!done[i] && (plan->nodes[i].chain == main_chain) == want_main && left[i] == 0) return (int)i;
                return -1;
            };
            while ((int)neworder.size() < N) {
                int m = next_ready(true, scan_main), sd = next_ready(false, scan_side);
                int pick = (m >= 0 && (run < burst || sd < 0)) ? m : sd;
                if (pick < 0) pick = m;
                if (pick < 0) { delete plan; return PDGN_ERR_INVALID; }
                run = (pick == m) ? run + 1 : 0;
                done[pick] = 1;
                neworder.push_back(pick);
                for (int v : psucc[pick]) --left[v];
            }
            std::vector<RNode> nn;
            nn.reserve(N);
            for (int i : neworder) nn.push_back(plan->nodes[i]);
            plan->nodes.swap(nn);
        }
    }
    {
        // is every chain's tail an ancestor of the issuing chain's tail?  (backward walk over the captured dependencies)
        int main_chain = -1;
        for (size_t c = 0; c < plan->chain_label.size(); ++c) if (plan->chain_label[c] == 0) main_chain = (int)c;
        if (main_chain >= 0) {
            std::vector<char> anc(n, 0);
            std::vector<int> stack{tail[main_chain]};
            anc[tail[main_chain]] = 1;
            while (!stack.empty()) {
                const int u = stack.back(); stack.pop_back();
                for (int d : deps[u]) if (!anc[d]) { anc[d] = 1; stack.push_back(d); }
            }
            plan->joined = 1;
            for (size_t c = 0; c < tail.size(); ++c) if (!anc[tail[c]]) plan->joined = 0;
        }
    }
    for (auto &ev : plan->events)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) { delete plan; return (int)e; }
    plan->chain_stream.assign(tail.size(), nullptr);
    plan->counts[0] = (int)n;
    plan->counts[5] = (int)tail.size();
    plan->counts[6] = (int)plan->events.size();
    for (int l : plan->chain_label) if (l >= 0) ++plan->counts[7];
    *plan_out = plan;
    return 0;
}

extern "C" int pdgn_replay_info(void *plan_, int *counts8) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || !counts8) return PDGN_ERR_INVALID;
    memcpy(counts8, plan->counts, sizeof(plan->counts));
    return 0;
}

// labels[c] = marker id of chain c (or -1), nodes_per_chain[c] = its length; both arrays of counts[5] ints.
extern "C" int pdgn_replay_chains(void *plan_, int *labels, int *nodes_per_chain) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan) return PDGN_ERR_INVALID;
    size_t nc = plan->chain_label.size();
    for (size_t c = 0; c < nc; ++c) { if (labels) labels[c] = plan->chain_label[c]; if (nodes_per_chain) nodes_per_chain[c] = 0; }
    if (nodes_per_chain) for (auto &r : plan->nodes) ++nodes_per_chain[r.chain];
    return 0;
}

extern "C" int pdgn_replay_set_stream(void *plan_, int chain, pdgn_stream_t stream) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || chain < 0 || chain >= (int)plan->chain_stream.size()) return PDGN_ERR_INVALID;
    plan->chain_stream[chain] = (hipStream_t)stream;
    return 0;
}

// Nodes [lo, hi) of the list, in order (the whole list: 0 .. counts[0]).  A caller that issues the list in two ranges can
// record an event of its own between them (pdgn_amd/trainer.py paces the next iteration's issue on one).
extern "C" int pdgn_replay_launch_range(void *plan_, int lo, int hi) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || lo < 0 || hi > (int)plan->nodes.size() || lo > hi) return PDGN_ERR_INVALID;
    hipError_t e = hipSuccess;
    for (int i = lo; i < hi; ++i) {
        RNode &r = plan->nodes[i];
        hipStream_t s = plan->chain_stream[r.chain];
        for (int w : r.waits)
            if ((e = hipStreamWaitEvent(s, plan->events[w], 0)) != hipSuccess) return (int)e;
        if (r.tstart >= 0 && plan->time_count[r.tstart] < plan->time_slots &&
            (e = hipEventRecord(plan->time_ev[(r.tstart * plan->time_slots + plan->time_count[r.tstart]) * 2], s)) != hipSuccess)
            return (int)e;
        switch (r.kind) {
        case NK_KERNEL:
            e = hipModuleLaunchKernel(r.func, r.gx, r.gy, r.gz, r.bx, r.by, r.bz, r.shmem, s, r.params, r.extra);
            break;
        case NK_MEMSET:
            if (r.elem == 4) e = hipMemsetD32Async((hipDeviceptr_t)r.dst, (int)r.value, r.width, s);
            else if (r.elem == 2) e = hipMemsetD16Async((hipDeviceptr_t)r.dst, (unsigned short)r.value, r.width, s);
            else e = hipMemsetAsync(r.dst, (int)r.value, r.width, s);
            break;
        case NK_MEMCPY:
            e = hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyDeviceToDevice, s);
            break;
        default:
            break;
        }
        if (e != hipSuccess) return (int)e;
        if (r.tstop >= 0 && plan->time_count[r.tstop] < plan->time_slots) {
            if ((e = hipEventRecord(plan->time_ev[(r.tstop * plan->time_slots + plan->time_count[r.tstop]) * 2 + 1], s)) != hipSuccess)
                return (int)e;
            ++plan->time_count[r.tstop];
        }
        if (r.record >= 0 && (e = hipEventRecord(plan->events[r.record], s)) != hipSuccess) return (int)e;
    }
    return 0;
}

// Kernel nodes launched through host symbol `sym` with grid.x == gx (gx < 0: any grid): their list positions (at most max_out
// written); returns their number.  `sym` identifies a kernel INSTANCE (e.g. pdgn_gemm_nt_ps_launch_info).
extern "C" int pdgn_replay_kernel_nodes(void *plan_, const void *sym, int gx, int *pos, int max_out) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || !sym || max_out < 0) return PDGN_ERR_INVALID;
    int n = 0;
    for (size_t i = 0; i < plan->nodes.size(); ++i) {
        const RNode &r = plan->nodes[i];
        if (r.kind == NK_KERNEL && r.sym == sym && (gx < 0 || (int)r.gx == gx)) {
            if (n < max_out && pos) pos[n] = (int)i;
            ++n;
        }
    }
    return n;
}

// List position of the next (dir > 0) / previous (dir < 0) node of the same chain as the node at `pos`, or -1; kind_out (may be
// NULL) receives that node's kind (0 kernel, 1 memset, 2 memcpy, 3 empty) and sym_out its kernel symbol.
extern "C" int pdgn_replay_chain_neighbor(void *plan_, int pos, int dir, int *kind_out, const void **sym_out) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || pos < 0 || pos >= (int)plan->nodes.size() || dir == 0) return -1;
    const int c = plan->nodes[pos].chain, n = (int)plan->nodes.size();
    for (int i = pos + (dir > 0 ? 1 : -1); i >= 0 && i < n; i += (dir > 0 ? 1 : -1))
        if (plan->nodes[i].chain == c) {
            if (kind_out) *kind_out = plan->nodes[i].kind;
            if (sym_out) *sym_out = plan->nodes[i].sym;
            return i;
        }
    return -1;
}

// Time SPANS of the list INSIDE the iterations that follow: span i runs from in front of the node at first[i] to behind the node at
// last[i] (both of one chain, i.e. one stream): two timing events on that stream per pass (bench.py: "measured live ... with HIP
// events over the timed region, on the stream the kernel is launched on"), for the next `slots` passes.  pdgn_replay_time_read
// returns the durations (ms_out[i * slots + j], j < counts[i]) once they have completed; n = 0 switches the timing off.
extern "C" int pdgn_replay_time_spans(void *plan_, const int *first, const int *last, int n, int slots) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || n < 0 || slots < 0 || (n > 0 && (!first || !last))) return PDGN_ERR_INVALID;
    // (validate every span and build the events BEFORE anything of the plan changes -- a refused call, or an event that
    // cannot be created, leaves the plan with no timing at all, never with tags that index a missing event)
    const int N = (int)plan->nodes.size();
    for (int i = 0; i < n; ++i)
        if (first[i] < 0 || last[i] >= N || first[i] > last[i] || plan->nodes[first[i]].chain != plan->nodes[last[i]].chain)
            return PDGN_ERR_INVALID;
    std::vector<hipEvent_t> events((size_t)n * slots * 2, nullptr);
    for (auto &ev : events) {
        const hipError_t e = hipEventCreate(&ev);
        if (e != hipSuccess) {
            for (auto made : events) if (made) (void)hipEventDestroy(made);
            return (int)e;
        }
    }
    for (auto &r : plan->nodes) r.tstart = r.tstop = -1;
    for (auto ev : plan->time_ev) if (ev) (void)hipEventDestroy(ev);
    plan->time_ev.swap(events);
    plan->time_count.assign((size_t)n, 0);
    plan->time_slots = n ? slots : 0;
    for (int i = 0; i < n; ++i) {
        plan->nodes[first[i]].tstart = i;
        plan->nodes[last[i]].tstop = i;
    }
    return 0;
}

extern "C" int pdgn_replay_time_read(void *plan_, float *ms_out, int *counts) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || !ms_out || !counts) return PDGN_ERR_INVALID;
    for (size_t i = 0; i < plan->time_count.size(); ++i) {
        counts[i] = plan->time_count[i];
        for (int j = 0; j < plan->time_count[i]; ++j) {
            const size_t k = (i * plan->time_slots + j) * 2;
            hipError_t e = hipEventSynchronize(plan->time_ev[k + 1]);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms_out[i * plan->time_slots + j], plan->time_ev[k], plan->time_ev[k + 1]);
            if (e != hipSuccess) return (int)e;
        }
    }
    return 0;
}

extern "C" int pdgn_replay_launch(void *plan_) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan) return PDGN_ERR_INVALID;
    return pdgn_replay_launch_range(plan_, 0, (int)plan->nodes.size());
}

// Host points of the list (markers with id >= 64), in list order: ids[i], list position pos[i], chain[i]; returns their number
// (at most max_out are written), or a negative error.
extern "C" int pdgn_replay_points(void *plan_, int *ids, int *pos, int *chain, int max_out) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || max_out < 0) return PDGN_ERR_INVALID;
    int n = 0;
    for (size_t i = 0; i < plan->nodes.size(); ++i)
        if (plan->nodes[i].point >= 0) {
            if (n < max_out) {
                if (ids) ids[n] = plan->nodes[i].point;
                if (pos) pos[n] = (int)i;
                if (chain) chain[n] = plan->nodes[i].chain;
            }
            ++n;
        }
    return n;
}

// 1 when the issuing chain's last node comes after every chain's last node in the captured dependencies, else 0.
extern "C" int pdgn_replay_joined(void *plan_) {
    RPlan *plan = (RPlan *)plan_;
    return plan ? plan->joined : 0;
}

// Position in the list of the n-th node (0-based) of chain `chain`, or -1: lets a caller cut the list at a point of one
// stream's progress (e.g. "after 60 % of the issuing stream's launches").
extern "C" int pdgn_replay_position(void *plan_, int chain, int nth) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || nth < 0) return -1;
    int seen = 0;
    for (size_t i = 0; i < plan->nodes.size(); ++i)
        if (plan->nodes[i].chain == chain && seen++ == nth) return (int)i;
    return -1;
}

// Measurement only: pdgn_replay_launch with the host time of every call accumulated per kind -- us[0..3] = kernel launches,
// memsets + copies, event waits, event records; us[4 + c] = everything issued for chain c (c < 28).
extern "C" int pdgn_replay_launch_timed(void *plan_, double *us32) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || !us32) return PDGN_ERR_INVALID;
    for (int i = 0; i < 32; ++i) us32[i] = 0.0;
    using clk = std::chrono::steady_clock;
    auto since = [](clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); };
    hipError_t e = hipSuccess;
    for (RNode &r : plan->nodes) {
        hipStream_t s = plan->chain_stream[r.chain];
        auto t0 = clk::now();
        for (int w : r.waits)
            if ((e = hipStreamWaitEvent(s, plan->events[w], 0)) != hipSuccess) return (int)e;
        double tw = since(t0);
        auto t1 = clk::now();
        int slot = 1;
        switch (r.kind) {
        case NK_KERNEL:
            e = hipModuleLaunchKernel(r.func, r.gx, r.gy, r.gz, r.bx, r.by, r.bz, r.shmem, s, r.params, r.extra);
            slot = 0;
            break;
        case NK_MEMSET:
            if (r.elem == 4) e = hipMemsetD32Async((hipDeviceptr_t)r.dst, (int)r.value, r.width, s);
            else if (r.elem == 2) e = hipMemsetD16Async((hipDeviceptr_t)r.dst, (unsigned short)r.value, r.width, s);
            else e = hipMemsetAsync(r.dst, (int)r.value, r.width, s);
            break;
        case NK_MEMCPY:
            e = hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyDeviceToDevice, s);
            break;
        default:
            break;
        }
        if (e != hipSuccess) return (int)e;
        double tl = since(t1);
        auto t2 = clk::now();
        if (r.record >= 0 && (e = hipEventRecord(plan->events[r.record], s)) != hipSuccess) return (int)e;
        double tr = since(t2);
        us32[slot] += tl; us32[2] += tw; us32[3] += tr;
        if (r.chain < 28) us32[4 + r.chain] += tl + tw + tr;
    }
    return 0;
}

// Measurement only: one replay with a timing event recorded after every `stride`-th node of chain `chain` (and one before its
// first node); after a device synchronise, ms_out[i] = milliseconds from the first event to the i-th, pos_out[i] = how many
// nodes of the chain had been issued by then.  Returns the number of probes written (<= max_out), or a negative / HIP code.
extern "C" int pdgn_replay_probe_chain(void *plan_, int chain, int stride, float *ms_out, int *pos_out, int max_out) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan || stride < 1 || !ms_out || !pos_out || max_out < 2 || chain < 0 || chain >= (int)plan->chain_stream.size())
        return PDGN_ERR_INVALID;
    std::vector<hipEvent_t> ev;
    std::vector<int> at;
    hipError_t e = hipSuccess;
    hipStream_t cs = plan->chain_stream[chain];
    auto probe = [&](int count) {
        if ((int)ev.size() >= max_out) return;
        hipEvent_t x;
        if (hipEventCreate(&x) != hipSuccess) return;
        (void)hipEventRecord(x, cs);
        ev.push_back(x);
        at.push_back(count);
    };
    int seen = 0;
    bool first = true;
    for (RNode &r : plan->nodes) {
        hipStream_t s = plan->chain_stream[r.chain];
        for (int w : r.waits)
            if ((e = hipStreamWaitEvent(s, plan->events[w], 0)) != hipSuccess) return (int)e;
        if (r.chain == chain && first) { probe(0); first = false; }
        switch (r.kind) {
        case NK_KERNEL:
            e = hipModuleLaunchKernel(r.func, r.gx, r.gy, r.gz, r.bx, r.by, r.bz, r.shmem, s, r.params, r.extra);
            break;
        case NK_MEMSET:
            if (r.elem == 4) e = hipMemsetD32Async((hipDeviceptr_t)r.dst, (int)r.value, r.width, s);
            else if (r.elem == 2) e = hipMemsetD16Async((hipDeviceptr_t)r.dst, (unsigned short)r.value, r.width, s);
            else e = hipMemsetAsync(r.dst, (int)r.value, r.width, s);
            break;
        case NK_MEMCPY:
            e = hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyDeviceToDevice, s);
            break;
        default:
            break;
        }
        if (e != hipSuccess) return (int)e;
        if (r.record >= 0 && (e = hipEventRecord(plan->events[r.record], s)) != hipSuccess) return (int)e;
        if (r.chain == chain && (++seen % stride) == 0) probe(seen);
    }
    probe(seen);
    if ((e = hipDeviceSynchronize()) != hipSuccess) return (int)e;
    for (size_t i = 0; i < ev.size(); ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[0], ev[i]);
        ms_out[i] = ms;
        pos_out[i] = at[i];
    }
    for (hipEvent_t x : ev) (void)hipEventDestroy(x);
    return (int)ev.size();
}

extern "C" int pdgn_replay_destroy(void *plan_) {
    RPlan *plan = (RPlan *)plan_;
    if (!plan) return 0;
    for (auto ev : plan->events) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : plan->time_ev) if (ev) (void)hipEventDestroy(ev);
    delete plan;
    return 0;
}
