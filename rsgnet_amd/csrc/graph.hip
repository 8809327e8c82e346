// Explicit HIP-graph construction for the forward launch list (gfx950 / ROCm 7.2).
//
// The plan records ~160 launches per forward with their data dependencies.  Replaying them from the
// host costs ~12 us of Python per launch (longer than many of the kernels) and stream capture of the
// multi-stream schedule crashes inside hipStreamEndCapture on this ROCm, so the graph is built node by
// node instead: while a graph is "open" on the calling thread every rsg_* launch becomes a kernel node
// whose dependencies are the node ids announced with rsg_graph_next_deps().  Independent HRNet
// branches are then free to run concurrently, and dependent nodes follow each other without the host.
#include <vector>
#include "common.h"

struct RsgGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipGraphNode_t> nodes;
    std::vector<hipGraphNode_t> deps;
    size_t op_begin = 0;     // nodes.size() when the current op's dependencies were announced
};

static thread_local RsgGraph* tl_open = nullptr;
thread_local int g_rsg_launch_status = 0;

int rsg_launch_raw(const void* fn, dim3 grid, dim3 block, unsigned lds, hipStream_t stream, void** params) {
    if (tl_open) {
        hipKernelNodeParams np;
        memset(&np, 0, sizeof(np));
        np.func = const_cast<void*>(fn);
        np.gridDim = grid;
        np.blockDim = block;
        np.sharedMemBytes = lds;
        np.kernelParams = params;
        np.extra = nullptr;
        hipGraphNode_t node;
        // second and later launches of one op (e.g. rsg_trp_tail) follow the op's previous node
        if (tl_open->nodes.size() > tl_open->op_begin) {
            tl_open->deps.clear();
            tl_open->deps.push_back(tl_open->nodes.back());
        }
        hipError_t e = hipGraphAddKernelNode(&node, tl_open->graph, tl_open->deps.data(), tl_open->deps.size(), &np);
        tl_open->deps.clear();
        if (e != hipSuccess) {
            snprintf(g_rsg_err, sizeof(g_rsg_err), "hipGraphAddKernelNode: %s", hipGetErrorString(e));
            return g_rsg_launch_status = 1;
        }
        tl_open->nodes.push_back(node);
        return g_rsg_launch_status = 0;
    }
    hipError_t e = hipLaunchKernel(fn, grid, block, params, lds, stream);
    if (e != hipSuccess) {
        snprintf(g_rsg_err, sizeof(g_rsg_err), "hipLaunchKernel: %s", hipGetErrorString(e));
        return g_rsg_launch_status = 1;
    }
    return g_rsg_launch_status = 0;
}

extern "C" int rsg_graph_begin(void** out) {
    if (!out) RSG_FAIL("graph_begin: null pointer");
    if (tl_open) RSG_FAIL("graph_begin: a graph is already open on this thread");
    RsgGraph* g = new RsgGraph();
    if (hipGraphCreate(&g->graph, 0) != hipSuccess) {
        delete g;
        RSG_FAIL("graph_begin: hipGraphCreate failed");
    }
    tl_open = g;
    *out = g;
    return 0;
}

extern "C" int rsg_graph_next_deps(void* graph, const int32_t* node_ids, int32_t n) {
    RsgGraph* g = (RsgGraph*)graph;
    if (!g || g != tl_open) RSG_FAIL("graph_next_deps: graph is not open on this thread");
    g->deps.clear();
    g->op_begin = g->nodes.size();
    for (int i = 0; i < n; ++i) {
        if (node_ids[i] < 0 || (size_t)node_ids[i] >= g->nodes.size())
            RSG_FAIL("graph_next_deps: node id %d out of range (have %zu nodes)", node_ids[i], g->nodes.size());
        g->deps.push_back(g->nodes[node_ids[i]]);
    }
    return 0;
}

extern "C" int rsg_graph_node_count(void* graph) {
    RsgGraph* g = (RsgGraph*)graph;
    return g ? (int)g->nodes.size() : -1;
}

extern "C" int rsg_graph_end(void* graph) {
    RsgGraph* g = (RsgGraph*)graph;
    if (!g || g != tl_open) RSG_FAIL("graph_end: graph is not open on this thread");
    tl_open = nullptr;
    hipError_t e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) RSG_FAIL("graph_end: hipGraphInstantiate: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rsg_graph_abort(void* graph) {
    if (graph && graph == tl_open) tl_open = nullptr;
    return 0;
}

extern "C" int rsg_graph_launch(void* graph, void* stream) {
    RsgGraph* g = (RsgGraph*)graph;
    if (!g || !g->exec) RSG_FAIL("graph_launch: graph was not instantiated");
    hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
    if (e != hipSuccess) RSG_FAIL("graph_launch: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rsg_graph_destroy(void* graph) {
    RsgGraph* g = (RsgGraph*)graph;
    if (!g) return 0;
    if (g == tl_open) tl_open = nullptr;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return 0;
}
