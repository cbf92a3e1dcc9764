// lpx_graph.hpp -- replay of launch-bound iteration chains as hipGraphs.
//
// An iteration is a fixed chain of kernels whose arguments (device pointers, sizes) never change for a given
// solver state, and all data-dependent control lives in the device control block.  So U iterations can be
// captured ONCE into a hipGraph and replayed: the host cost per iteration drops from ~4 us per launch to the
// graph's amortised launch, and kernels follow each other at the ~1.5 us boundary cost.
// Graphs are skipped for short requests, while K1 event timing is on, or with LPX_GRAPH=0.
#pragma once
#include <string.h>

#include <mutex>
#include <utility>
#include <vector>

#include "lpx_device.hpp"
#include "lpx_rank1.hpp"

struct LpxGraphEntry {
    std::vector<char> key;
    hipGraphExec_t exec;
};
struct LpxGraphCache {
    std::mutex mu;
    std::vector<LpxGraphEntry> entries;
    std::vector<std::pair<int, hipStream_t>> capture_streams; // one per device that ever captured
};
LpxGraphCache &lpx_graph_cache();

static inline int lpx_graph_unroll()
{
    // always even: the single-launch tableau chain alternates two buffers with the parity of the iteration, and a
    // replayed graph must end on the parity it started from
    int u = lpx_env_int("LPX_GRAPH_UNROLL", 16);
    u = u < 2 ? 2 : (u > 256 ? 256 : u);
    return u + (u & 1);
}

// enqueue_one(stream) enqueues exactly one iteration.  Returns LPX_OK or an error code.
template <class F>
static int lpx_run_iterations(const void *key, size_t keylen, int64_t iters, hipStream_t st, F enqueue_one)
{
    const int U = lpx_graph_unroll();
    const bool use_graph = lpx_env_int("LPX_GRAPH", 1) != 0 && !lpx_k1_profile().enabled && iters >= 2 * (int64_t)U;
    if (use_graph) {
        LpxGraphCache &gc = lpx_graph_cache();
        std::lock_guard<std::mutex> lock(gc.mu);
        int dev = 0;
        LPX_HIP(hipGetDevice(&dev));
        std::vector<char> k((const char *)key, (const char *)key + keylen);
        k.push_back((char)U);
        k.push_back((char)dev);
        hipGraphExec_t exec = nullptr;
        for (auto &e : gc.entries)
            if (e.key == k) exec = e.exec;
        if (!exec) {
            hipStream_t cs = nullptr;
            for (auto &p : gc.capture_streams)
                if (p.first == dev) cs = p.second;
            if (!cs) {
                LPX_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                gc.capture_streams.push_back({dev, cs});
            }
            hipGraph_t graph = nullptr;
            LPX_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
            for (int u = 0; u < U; ++u) enqueue_one(cs);
            LPX_HIP(hipStreamEndCapture(cs, &graph));
            LPX_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            LPX_HIP(hipGraphDestroy(graph));
            if (gc.entries.size() >= 32) {
                (void)hipGraphExecDestroy(gc.entries.front().exec);
                gc.entries.erase(gc.entries.begin());
            }
            gc.entries.push_back({k, exec});
        }
        while (iters >= U) {
            LPX_HIP(hipGraphLaunch(exec, st));
            iters -= U;
        }
    }
    for (int64_t it = 0; it < iters; ++it) enqueue_one(st);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
