// Error reporting and hipGraph capture helpers of libmg_hip.so.
#include <stdarg.h>
#include <string.h>

#include "mg_common.h"

static thread_local char g_err[512] = "";

void mg_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *mg_last_error(void) { return g_err; }
extern "C" int mg_abi_version(void) { return MG_ABI_VERSION; }

// Everything enqueued on `stream` between mg_graph_begin and mg_graph_end becomes one executable
// graph; replaying it costs one launch instead of one per kernel, which is what matters for the
// coarse levels of the V-cycle where each kernel runs for a few microseconds.
extern "C" int mg_graph_begin(void *stream)
{
    MG_REQUIRE(stream, "mg_graph_begin: the null stream cannot be captured");
    MG_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}

extern "C" int mg_graph_end(void *stream, void **graph_exec_out)
{
    MG_REQUIRE(stream && graph_exec_out, "mg_graph_end: null argument");
    hipGraph_t g = nullptr;
    MG_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        mg_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return 2;
    }
    *graph_exec_out = (void *)ge;
    return 0;
}

extern "C" int mg_graph_launch(void *graph_exec, void *stream)
{
    MG_REQUIRE(graph_exec, "mg_graph_launch: null graph");
    MG_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}

extern "C" int mg_graph_destroy(void *graph_exec)
{
    if (graph_exec) MG_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}
