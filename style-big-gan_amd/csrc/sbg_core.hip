// sbg_core.hip -- library-wide state of libsbg_hip.so: version and the thread-local error slot.
#include "sbg_common.h"

std::string& sbg_err_slot()
{
    static thread_local std::string slot;
    return slot;
}

extern "C" int sbg_version(void) { return 2; }

bool sbg_launch_geometry_ok(dim3 grid, dim3 block, size_t lds_bytes, const char* kernel, const char* file, int line)
{
    const uint64_t threads = (uint64_t)block.x * block.y * block.z;
    const bool ok = grid.x > 0 && grid.y > 0 && grid.z > 0 && grid.y <= 65535u && grid.z <= 65535u && block.x > 0 && block.y > 0 && block.z > 0
                 && threads <= 1024 && lds_bytes <= 160u * 1024u
                 && (uint64_t)grid.x * block.x <= 0xffffffffull;      // the dispatch packet's grid_size fields are 32-bit work-item counts
    if (!ok)
        (void)sbg_fail(SBG_ERR_INVALID, "%s:%d: refusing to launch %s with grid (%u, %u, %u), block (%u, %u, %u), %zu bytes of LDS",
                       file, line, kernel, grid.x, grid.y, grid.z, block.x, block.y, block.z, lds_bytes);
    return ok;
}

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

const char* sbg_env(const char* name)
{
    static std::mutex mu;
    static std::map<std::string, std::pair<bool, std::string>> seen;
    std::lock_guard<std::mutex> lk(mu);
    auto it = seen.find(name);
    if (it == seen.end()) {
        const char* v = getenv(name);
        it = seen.emplace(name, std::make_pair(v != nullptr, std::string(v ? v : ""))).first;
    }
    return it->second.first ? it->second.second.c_str() : nullptr;
}

extern "C" const char* sbg_last_error(void) { return sbg_err_slot().c_str(); }

// Experiment word: kernel variants that are A/B-tested in ONE process (interleaved rounds on one device: scratch/kbench.py) select on its bits.
// Initial value from SBG_EXPERIMENT; sbg_experiment_set returns the previous value.  Not part of the product's contract.
static std::atomic<int>& sbg_experiment_word()
{
    static std::atomic<int> w{[] { const char* e = getenv("SBG_EXPERIMENT"); return e ? atoi(e) : 0; }()};
    return w;
}
int sbg_experiment() { return sbg_experiment_word().load(std::memory_order_relaxed); }
extern "C" int sbg_experiment_set(int v) { return sbg_experiment_word().exchange(v); }

// ------------------------------------------------------------------------------------------------
// Launch timing log.
#include <vector>
#include <atomic>

namespace {
struct ProfEntry { sbg_prof_record rec; hipEvent_t start, stop; bool closed; };
std::mutex g_prof_mu;
std::vector<ProfEntry> g_prof_log;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
std::atomic<int> g_prof_enabled{0};
const size_t kProfMax = 1 << 20;
}

bool sbg_prof_on() { return g_prof_enabled.load(std::memory_order_relaxed) != 0; }

int sbg_prof_open(hipStream_t s, int kind, double flops, double bytes, const int* dims, int ndims)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_log.size() >= kProfMax) return -1;
    ProfEntry e;
    e.rec.kind = kind; e.rec.flops = flops; e.rec.bytes = bytes; e.rec.ms = 0.f; e.rec.pad = 0;
    for (int i = 0; i < 7; i++) e.rec.dims[i] = i < ndims ? dims[i] : 0;
    if (!g_prof_pool.empty()) { e.start = g_prof_pool.back().first; e.stop = g_prof_pool.back().second; g_prof_pool.pop_back(); }
    else {
        if (hipEventCreate(&e.start) != hipSuccess) return -1;
        if (hipEventCreate(&e.stop) != hipSuccess) { (void)hipEventDestroy(e.start); return -1; }
    }
    e.closed = false;
    (void)hipEventRecord(e.start, s);
    g_prof_log.push_back(e);
    return (int)g_prof_log.size() - 1;
}

void sbg_prof_close(hipStream_t s, int slot)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < 0 || slot >= (int)g_prof_log.size()) return;
    (void)hipEventRecord(g_prof_log[slot].stop, s);
    g_prof_log[slot].closed = true;
}

extern "C" int sbg_prof_enable(int on)
{
    return g_prof_enabled.exchange(on ? 1 : 0);
}

extern "C" int sbg_prof_fetch(sbg_prof_record* out, int max)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (out == nullptr) return (int)g_prof_log.size();
    int n = 0;
    for (auto& e : g_prof_log) {
        if (e.closed) {
            (void)hipEventSynchronize(e.stop);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) e.rec.ms = ms;
        }
        if (n < max) out[n++] = e.rec;
        g_prof_pool.emplace_back(e.start, e.stop);
    }
    g_prof_log.clear();
    return n;
}
