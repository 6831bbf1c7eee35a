// Host thread budget of the library (the reference's rayon pool size: std::thread::available_parallelism, main.rs:40-57,
// utils.rs:139-158).  A GPU box shows 256 logical CPUs to std::thread::hardware_concurrency() while its cgroup grants 16: more
// threads than the quota only get throttled, and eight ranks (or handles) that each start 16 of them are the first thing that
// would bend a weak-scaling curve.  Every worker pool of the library is sized by host_threads(): the CPUs this process may use
// (affinity mask capped by the cgroup quota) divided by the number of ranks/handles that share the host.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "rtx_internal.hpp"

namespace rtx {

static unsigned cgroup_quota_cpus() {  // 0: no quota
    unsigned out = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[64];
        unsigned long long period = 0;
        if (fscanf(f, "%63s %llu", q, &period) == 2 && q[0] != 'm' && period > 0) {
            const unsigned long long quota = strtoull(q, nullptr, 10);
            out = (unsigned)std::max<unsigned long long>(1, quota / period);
        }
        fclose(f);
        return out;
    }
    long long quota = -1, period = 0;  // cgroup v1
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lld", &quota) != 1) quota = -1; fclose(f); }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lld", &period) != 1) period = 0; fclose(f); }
    if (quota > 0 && period > 0) out = (unsigned)std::max<long long>(1, quota / period);
    return out;
}

unsigned available_parallelism() {
    static const unsigned n = [] {
        unsigned cpus = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = (unsigned)CPU_COUNT(&set);
        if (cpus == 0) cpus = std::max(1u, std::thread::hardware_concurrency());
        const unsigned quota = cgroup_quota_cpus();
        if (quota) cpus = std::min(cpus, quota);
        return std::max(1u, cpus);
    }();
    return n;
}

// ranks / handles that share this host's CPUs: rtx_set_host_share, else LOCAL_WORLD_SIZE (what torch.distributed.run exports), else 1
static std::atomic<unsigned> g_host_share{0};

unsigned host_share() {
    unsigned s = g_host_share.load(std::memory_order_relaxed);
    if (s == 0) {
        const char *e = getenv("LOCAL_WORLD_SIZE");
        const long v = e ? strtol(e, nullptr, 10) : 1;
        s = v >= 1 && v <= 4096 ? (unsigned)v : 1u;
        g_host_share.store(s, std::memory_order_relaxed);
    }
    return s;
}

unsigned host_threads(unsigned want, unsigned sharers) {
    const unsigned budget = std::max(1u, available_parallelism() / std::max(1u, host_share() * std::max(1u, sharers)));
    return std::max(1u, std::min(want, budget));
}

// Hardware queues of the HIP runtime (GPU_MAX_HW_QUEUES, 4 by default, read when the runtime initialises): a handle drives five streams -- two for
// kernels, two for transfers, the runtime's own -- and streams beyond the queues SHARE one: a D2H copy then waits behind every kernel that was
// enqueued on its queue-mate before it.  With two chunks of rtx_raxtax on the device (RTX_OPT_RUN_AHEAD) that was the whole front half of the
// next chunk: 113 ms per 1 M queries where six queues or more give 86 (four queues without the run-ahead: 92).  The library asks for eight when
// it is loaded, unless the process has set the variable itself; a host that initialised HIP before loading the library keeps what it had -- it
// can export the variable itself -- and rtx_raxtax leaves the run-ahead off unless the variable reads six or more.
static const int g_hw_queues_asked = [] {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    return 0;
}();
bool hw_queues_for_run_ahead() {
    (void)g_hw_queues_asked;
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    return e && strtol(e, nullptr, 10) >= 6;
}

}  // namespace rtx

extern "C" int rtx_set_host_share(uint32_t n_ranks_on_this_host) {
    if (n_ranks_on_this_host == 0 || n_ranks_on_this_host > 4096) { rtx::set_error("rtx_set_host_share: %u ranks", n_ranks_on_this_host); return RTX_ERR_INVALID; }
    rtx::g_host_share.store(n_ranks_on_this_host, std::memory_order_relaxed);
    return RTX_OK;
}
extern "C" uint32_t rtx_host_threads(void) { return rtx::host_threads(~0u, 1); }
