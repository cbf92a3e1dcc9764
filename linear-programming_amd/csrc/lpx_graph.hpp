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
    hipEvent_t done; // recorded after every launch of exec: an evicted exec is destroyed only once its last launch retired
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

// enqueue_one may return void or an LPX_* code
template <class F>
static inline auto lpx_call_enqueue(F &f, hipStream_t st, int) -> decltype((int)f(st))
{
    return (int)f(st);
}
template <class F>
static inline int lpx_call_enqueue(F &f, hipStream_t st, long)
{
    f(st);
    return LPX_OK;
}

// Capture U iterations on the device's capture stream and instantiate them.  Whatever fails after BeginCapture, the
// capture is ended and the graph destroyed before returning, so the capture stream never stays in capture mode.
// Returns LPX_OK with *exec set; an error of enqueue_one (propagated: the caller's request is at fault); or LPX_E_HIP
// with *exec == nullptr when capture / instantiation itself failed (the caller then launches eagerly).
template <class F>
static int lpx_capture_iterations(hipStream_t cs, int U, F &enqueue_one, hipGraphExec_t *exec)
{
    *exec = nullptr;
    hipGraph_t graph = nullptr;
    LPX_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
    int rc = LPX_OK;
    const char *inject = getenv("LPX_GRAPH_TEST_FAIL"); // test hook: "arg" = an iteration reports LPX_E_ARG, "hip" = capture fails
    for (int u = 0; u < U && rc == LPX_OK; ++u) {
        rc = lpx_call_enqueue(enqueue_one, cs, 0);
        if (rc == LPX_OK && inject && inject[0] == 'a' && u == 1) {
            lpx_set_error("lpx_run_iterations: injected argument error inside a captured chain");
            rc = LPX_E_ARG;
        }
    }
    hipError_t e = hipGetLastError(); // a failed launch inside the capture
    const hipError_t e2 = hipStreamEndCapture(cs, &graph); // always: leaves capture mode even after a failure
    if (e == hipSuccess) e = e2;
    if (e == hipSuccess && inject && inject[0] == 'h') e = hipErrorUnknown;
    if (rc == LPX_OK && e == hipSuccess && graph != nullptr) {
        e = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) *exec = nullptr;
    }
    if (graph != nullptr) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    if (rc != LPX_OK) return rc;
    if (e != hipSuccess) {
        lpx_set_error("graph capture failed: %s (falling back to eager launches)", hipGetErrorString(e));
        return LPX_E_HIP;
    }
    return LPX_OK;
}

// enqueue_one(stream) enqueues exactly one iteration (returns void or an LPX_* code).  Returns LPX_OK or an error code.
// unroll > 0: iterations per graph fixed by the caller (a chain whose kernel arguments repeat with that period);
// whole_graphs: replay from iters >= unroll on, also while K1 event timing is on (the caller launches the timed kernel
// itself, outside the graph).
template <class F>
static int lpx_run_iterations(const void *key, size_t keylen, int64_t iters, hipStream_t st, F enqueue_one, int unroll = 0,
                              bool whole_graphs = false)
{
    const int U = unroll > 0 ? unroll : lpx_graph_unroll();
    const bool use_graph = lpx_env_int("LPX_GRAPH", 1) != 0 &&
                           (whole_graphs ? iters >= (int64_t)U : (!lpx_k1_profile().enabled && iters >= 2 * (int64_t)U));
    if (use_graph) {
        LpxGraphCache &gc = lpx_graph_cache();
        std::lock_guard<std::mutex> lock(gc.mu);
        int dev = 0;
        LPX_HIP(hipGetDevice(&dev));
        std::vector<char> k((const char *)key, (const char *)key + keylen);
        const int tail[2] = {U, dev}; // whole ints: U = 256 or device 256 must not alias 0
        k.insert(k.end(), (const char *)tail, (const char *)tail + sizeof tail);
        LpxGraphEntry *hit = nullptr;
        for (auto &e : gc.entries)
            if (e.key == k) hit = &e;
        if (!hit) {
            hipStream_t cs = nullptr;
            for (auto &p : gc.capture_streams)
                if (p.first == dev) cs = p.second;
            if (!cs) {
                LPX_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                gc.capture_streams.push_back({dev, cs});
            }
            hipGraphExec_t exec = nullptr;
            const int rc = lpx_capture_iterations(cs, U, enqueue_one, &exec);
            if (rc != LPX_OK && rc != LPX_E_HIP) return rc; // the iteration itself is at fault
            if (exec != nullptr) {
                hipEvent_t done = nullptr;
                if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGraphExecDestroy(exec);
                    exec = nullptr;
                } else {
                    if (gc.entries.size() >= 32) { // evict the oldest once its last launch has retired
                        (void)hipEventSynchronize(gc.entries.front().done);
                        (void)hipGraphExecDestroy(gc.entries.front().exec);
                        (void)hipEventDestroy(gc.entries.front().done);
                        gc.entries.erase(gc.entries.begin());
                    }
                    gc.entries.push_back({k, exec, done});
                    hit = &gc.entries.back();
                }
            }
        }
        if (hit) {
            while (iters >= U) {
                LPX_HIP(hipGraphLaunch(hit->exec, st));
                iters -= U;
            }
            LPX_HIP(hipEventRecord(hit->done, st));
        }
    }
    for (int64_t it = 0; it < iters; ++it) {
        const int rc = lpx_call_enqueue(enqueue_one, st, 0);
        if (rc != LPX_OK) return rc;
    }
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
