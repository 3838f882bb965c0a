// Error reporting and hipGraph capture helpers of libmg_hip.so.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

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

// Streams made by mg_graph_begin_own, which mg_graph_end releases.
static std::mutex g_own_mutex;
static std::vector<hipStream_t> g_own_streams;

// A capture on a stream that exists only for it.  In a process that holds an RCCL communicator torch's watchdog
// thread polls the end events of outstanding collectives; on this HIP runtime such a poll fails with
// hipErrorCapturedEvent (and torch ends the process) while a stream that has waited for one of those events is
// capturing.  A stream that has never carried anything cannot have; and it is gone again before the caller's next
// stream comes into use, so it does not shift the hardware queues the caller's streams are spread over.
extern "C" int mg_graph_begin_own(void **stream_out)
{
    MG_REQUIRE(stream_out, "mg_graph_begin_own: null argument");
    hipStream_t s = nullptr;
    MG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(s);
        mg_set_error("mg_graph_begin_own: hipStreamBeginCapture failed: %s", hipGetErrorString(e));
        return 2;
    }
    {
        std::lock_guard<std::mutex> lk(g_own_mutex);
        g_own_streams.push_back(s);
    }
    *stream_out = (void *)s;
    return 0;
}

static void release_if_own(hipStream_t s)
{
    bool own = false;
    {
        std::lock_guard<std::mutex> lk(g_own_mutex);
        auto it = std::find(g_own_streams.begin(), g_own_streams.end(), s);
        if (it != g_own_streams.end()) {
            g_own_streams.erase(it);
            own = true;
        }
    }
    if (own) (void)hipStreamDestroy(s);
}

extern "C" int mg_graph_end(void *stream, void **graph_exec_out)
{
    MG_REQUIRE(stream && graph_exec_out, "mg_graph_end: null argument");
    hipGraph_t g = nullptr;
    hipError_t ec = hipStreamEndCapture((hipStream_t)stream, &g);
    release_if_own((hipStream_t)stream);
    if (ec != hipSuccess) {
        mg_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ec));
        return 2;
    }
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
