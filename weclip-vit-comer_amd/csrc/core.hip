// Error reporting / version / device helpers of the weclip_hip C-ABI library.
#include "common.h"
#include <stdarg.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";

extern "C" void wc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* wc_last_error(void) { return g_err; }

extern "C" int wc_version(void) { return 100; }   // 0.1.0

// Number of visible HIP devices, or -1 (with wc_last_error set) when the runtime is unusable.
extern "C" int wc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        wc_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return -1;
    }
    return n;
}

// gcnArchName of device `dev` copied into buf (e.g. "gfx950:sramecc+:xnack-").
extern "C" int wc_device_arch(int dev, char* buf, int buflen) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        wc_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return WC_ERR_HIP;
    }
    strncpy(buf, p.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg): the launch sites of
// the dominant kernels bracket themselves with wc_prof_begin / wc_prof_end.  Off by default (one branch per launch).
// An event pair around a launch costs ~6 us of GPU time (it fences the neighbouring kernels), so bench.py records a
// regular 1-in-n sample of the instrumented launches rather than all ~290 per step (which costs 10 % throughput).
struct ProfRec { const char* name; const char* tag; hipEvent_t e0, e1; double work, bytes; int n; int weight; };
static int g_prof_stride = 0;          // 0 = off; n = record one of every n instrumented launches
static const char* g_prof_tag = "";    // group tag of the call site (e.g. "@vit_attn"), appended to the kernel name in the report
static unsigned long g_prof_ctr = 0;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t prof_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

extern "C" void wc_prof_enable(int stride) {
    for (auto& r : g_prof) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
    g_prof.clear();
    g_prof_stride = stride > 0 ? stride : 0;
    g_prof_ctr = 0;
}

// Group tag for the launches that follow (bench.py: "@vit_attn" around in-projection, attention, head-mean and
// out-projection of an encoder block).  tag must be a string that outlives the report (the Python host passes interned
// constants); NULL or "" clears it.
extern "C" void wc_prof_tag(const char* tag) {
    static std::vector<std::string> keep;
    if (!tag || !*tag) { g_prof_tag = ""; return; }
    for (auto& k : keep)
        if (k == tag) { g_prof_tag = k.c_str(); return; }
    keep.reserve(64);
    keep.push_back(tag);
    g_prof_tag = keep.back().c_str();
}

static int prof_begin_rec(void* stream, int weight);

// Every call records (while profiling is on): for brackets around MANY launches (one pair per PAR sweep group: two per step),
// where a 1-in-n pick would leave a handful of samples or none.  Such a record stands for itself (weight 1).
int wc_prof_begin_always(void* stream) {
    if (!g_prof_stride) return -1;
    return prof_begin_rec(stream, 1);
}

int wc_prof_begin(void* stream) {
    if (!g_prof_stride) return -1;
    // one of every `stride` instrumented launches ON AVERAGE, drawn by a fixed pseudo-random sequence: a regular 1-in-n pick
    // aliases with the step (252 instrumented launches per step and n = 7 sampled the same 36 launches in every step)
    g_prof_ctr = g_prof_ctr * 6364136223846793005UL + 1442695040888963407UL;
    if (g_prof_stride > 1 && (g_prof_ctr >> 33) % (unsigned long)g_prof_stride != 0) return -1;
    return prof_begin_rec(stream, g_prof_stride);       // stands for `stride` launches
}

static int prof_begin_rec(void* stream, int weight) {
    ProfRec r;
    r.weight = weight;
    r.name = "";
    r.tag = g_prof_tag;
    r.e0 = prof_event();
    r.e1 = prof_event();
    r.work = 0.0;
    r.bytes = 0.0;
    r.n = 1;
    hipEventRecord(r.e0, (hipStream_t)stream);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}

void wc_prof_end(int idx, const char* name, double work, void* stream) {
    if (idx < 0) return;
    g_prof[idx].name = name;          // string literals only
    g_prof[idx].work = work;
    hipEventRecord(g_prof[idx].e1, (hipStream_t)stream);
}

// The same with a second figure: the ALGORITHMIC HBM bytes of the launch, for kernels that are reported against both
// rooflines (the row-streaming GEMM: flops and bytes; the deformable-attention gathers: bytes only, work = 0).
void wc_prof_end2(int idx, const char* name, double work, double bytes, void* stream) {
    if (idx < 0) return;
    g_prof[idx].bytes = bytes;
    wc_prof_end(idx, name, work, stream);
}

// One event pair around n back-to-back launches of the same kernel (the PAR sweeps): `work` is the total of the n launches;
// the report counts n launches, so the average per launch carries the ~1 us between dependent kernels but not the ~6 us an
// event pair around every single launch adds by fencing it off from its neighbours.
void wc_prof_end_n(int idx, const char* name, double work, void* stream, int n) {
    if (idx < 0) return;
    g_prof[idx].n = n > 0 ? n : 1;
    wc_prof_end(idx, name, work, stream);
}

// Aggregated records after a device synchronisation: returns the number of distinct kernel names and writes
// "name\tlaunches\tms\twork\tms scaled by the sampling weight of each record\talgorithmic bytes\n" lines into buf (truncated at cap).
extern "C" int wc_prof_report(char* buf, int cap) {
    hipDeviceSynchronize();
    std::map<std::string, std::pair<double, std::pair<double, long>>> agg;   // name -> (ms, (work, launches))
    std::map<std::string, double> est;                                       // name -> sum of ms * record weight
    std::map<std::string, double> byt;                                       // name -> algorithmic bytes of the recorded launches
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        auto& a = agg[std::string(r.name) + r.tag];
        a.first += ms;
        a.second.first += r.work;
        a.second.second += r.n;
        est[std::string(r.name) + r.tag] += (double)ms * r.weight;
        byt[std::string(r.name) + r.tag] += r.bytes;
    }
    int off = 0;
    if (cap > 0) buf[0] = 0;
    for (auto& kv : agg) {
        const int n = snprintf(buf + off, off < cap ? cap - off : 0, "%s\t%ld\t%.6f\t%.6e\t%.6f\t%.6e\n", kv.first.c_str(),
                               kv.second.second.second, kv.second.first, kv.second.second.first, est[kv.first], byt[kv.first]);
        if (n < 0 || off + n >= cap) break;
        off += n;
    }
    return (int)agg.size();
}
