// node.hip -- the receivers of one node: N channels split into contiguous blocks over the node's GPUs.
//
// gnuais creates its receivers one by one and they share nothing (src/ais.c:141-147); the main loop hands every
// receiver the same interleaved buffer (src/ais.c:237-247).  A node object is that for many devices: device g owns
// channels [first_g, first_g + n_g) as one gnuais_batch of its own, every device has ONE host thread that issues
// its copies and launches (hipSetDevice is per thread) on the batch's own streams, and nothing is exchanged
// between devices -- no collective, no peer copy.  What comes back is merged on the host: frame records with
// GLOBAL channel numbers in the reference's order (channel, then time), counters per global channel.
//
// Built on the public batch ABI only (include/gnuais_hip.h); the HIP runtime is used for the host-buffer split
// (a strided 2-D copy per device) and nothing else.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gnuais_hip.h"

namespace {

thread_local std::string g_node_err;

int node_fail(int code, const std::string &what)
{
    g_node_err = what;
    return code;
}

double wall_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = GNUAIS_OK;
    std::string err;

    void loop()
    {
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return has_job || quit; });
                if (quit) return;
                j = std::move(job);
                has_job = false;
            }
            g_node_err.clear();
            const int r = j();
            // the job's own message (node_fail: hipSetDevice, staging, copies) if it left one, else the batch layer's
            // (both are thread-local and this is the thread the job ran on)
            std::string e = r == GNUAIS_OK ? std::string() : (!g_node_err.empty() ? g_node_err : std::string(gnuais_last_error()));
            {
                std::lock_guard<std::mutex> l(m);
                rc = r;
                err = std::move(e);
                done = true;
            }
            cv.notify_all();
        }
    }
    void submit(std::function<int()> j)
    {
        {
            std::lock_guard<std::mutex> l(m);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return done; });
        return rc;
    }
};

struct Shard {
    int device = 0, first = 0, n = 0;
    gnuais_batch *b = nullptr;
    int16_t *d_in[2] = {nullptr, nullptr};      // run_host: the shard's [len][n] slab, double-buffered
    hipStream_t s_in = nullptr;
    hipEvent_t e_free[2] = {nullptr, nullptr};  // the FIR that read d_in[q] is done (recorded behind the run on s_in)
    unsigned long long host_calls = 0;
    // where the shard's host thread runs: the NUMA node of its device (sysfs, via the device's PCI address) and how many
    // CPUs of that node the thread was pinned to (0: not pinned -- no sysfs entry, one node only, or GNUAIS_NODE_PIN=0)
    int numa_node = -1, pinned_cpus = 0;
    char pci[32] = {0};
    // since the last gnuais_node_mark(): calls, the host time spent inside the run calls (submission), and the wall clock
    // of the first submission and of the end of the shard's last sync -- so that one slow device shows by itself
    unsigned long long m_calls = 0;
    double m_submit_ms = 0.0, m_first = 0.0, m_last_sync = 0.0;
    Worker w;
};

// "0-15,32-47" -> CPUs set in `set`; returns how many
int parse_cpulist(const char *text, cpu_set_t *set)
{
    int n = 0;
    CPU_ZERO(set);
    for (const char *p = text; *p;) {
        char *e;
        long a = strtol(p, &e, 10), b = a;
        if (e == p) break;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int) c, set); ++n; }
        p = (*e == ',') ? e + 1 : e;
        if (*e != ',' ) break;
    }
    return n;
}

// Runs ON the shard's thread, before anything else: pin it to the CPUs of the NUMA node its device hangs off, so that
// what the thread allocates and touches from here on (the batch's pinned staging buffers, the launch path's queues)
// is local to that device's root complex.  Every step may fail quietly: the thread then stays where the OS put it.
void pin_to_device_node(Shard &s)
{
    if (hipDeviceGetPCIBusId(s.pci, (int) sizeof s.pci, s.device) != hipSuccess) { s.pci[0] = 0; return; }
    for (char *c = s.pci; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char) (*c - 'A' + 'a');   // sysfs spells it lower case
    char path[160], buf[4096];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", s.pci);
    FILE *f = fopen(path, "r");
    if (!f) return;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    s.numa_node = node;
    const char *off = getenv("GNUAIS_NODE_PIN");
    if (node < 0 || (off && atoi(off) == 0)) return;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return;
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!ok) return;
    cpu_set_t want, have, both;
    if (parse_cpulist(buf, &want) == 0 || sched_getaffinity(0, sizeof have, &have) != 0) return;
    CPU_AND(&both, &want, &have);               // only CPUs this process may use at all (cgroup / taskset)
    const int n = CPU_COUNT(&both);
    if (n > 0 && sched_setaffinity(0, sizeof both, &both) == 0) s.pinned_cpus = n;
}

} // namespace

struct gnuais_node {
    int N = 0, max_len = 0;
    std::vector<Shard *> shards;
    std::vector<gnuais_frame> scratch;
    std::string warnings;           // what create could not do without failing (one line per shard)
};

extern "C" {

const char *gnuais_node_last_error(void) { return g_node_err.c_str(); }

static int run_all(gnuais_node *nd, const std::function<int(Shard &)> &f)
{
    for (Shard *s : nd->shards) s->w.submit([s, &f] { return f(*s); });
    int rc = GNUAIS_OK;
    for (Shard *s : nd->shards) {
        const int r = s->w.wait();
        if (r != GNUAIS_OK && rc == GNUAIS_OK) {
            rc = r;
            g_node_err = "device " + std::to_string(s->device) + " (channels " + std::to_string(s->first) + ".." +
                         std::to_string(s->first + s->n - 1) + "): " + s->w.err;
        }
    }
    return rc;
}

void gnuais_node_destroy(gnuais_node *nd)
{
    if (!nd) return;
    for (Shard *s : nd->shards) {
        if (s->w.th.joinable()) {
            s->w.submit([s] {
                (void) hipSetDevice(s->device);
                if (s->b) gnuais_batch_destroy(s->b);
                for (int q = 0; q < 2; ++q) {
                    if (s->d_in[q]) (void) hipFree(s->d_in[q]);
                    if (s->e_free[q]) (void) hipEventDestroy(s->e_free[q]);
                }
                if (s->s_in) (void) hipStreamDestroy(s->s_in);
                return GNUAIS_OK;
            });
            s->w.wait();
            {
                std::lock_guard<std::mutex> l(s->w.m);
                s->w.quit = true;
            }
            s->w.cv.notify_all();
            s->w.th.join();
        }
        delete s;
    }
    delete nd;
}

int gnuais_node_create(gnuais_node **out, const int *devices, int n_devices, int n_channels, const float *taps,
                       int n_taps, unsigned pllinc, int max_len, int frame_capacity_per_device)
{
    if (!out) return node_fail(GNUAIS_E_ARG, "node_create: out is NULL");
    *out = nullptr;
    if (n_channels <= 0 || max_len <= 0) return node_fail(GNUAIS_E_ARG, "node_create: n_channels / max_len");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return node_fail(GNUAIS_E_HIP, "node_create: no HIP device");
    std::vector<int> devs;
    if (devices && n_devices > 0) devs.assign(devices, devices + n_devices);
    else for (int d = 0; d < ndev; ++d) devs.push_back(d);            // NULL / 0: every visible device
    for (int d : devs)
        if (d < 0 || d >= ndev) return node_fail(GNUAIS_E_ARG, "node_create: device index out of range");
    const int G = (int) std::min<size_t>(devs.size(), (size_t) n_channels);
    gnuais_node *nd = new gnuais_node;
    nd->N = n_channels;
    nd->max_len = max_len;
    for (int g = 0; g < G; ++g) {
        // contiguous blocks, SURVEY 8e: device g owns [g*N/G, (g+1)*N/G)
        const int lo = (int) ((long long) n_channels * g / G), hi = (int) ((long long) n_channels * (g + 1) / G);
        Shard *s = new Shard;
        s->device = devs[g];
        s->first = lo;
        s->n = hi - lo;
        s->w.th = std::thread([s] { pin_to_device_node(*s); s->w.loop(); });
        nd->shards.push_back(s);
    }
    const int rc = run_all(nd, [&](Shard &s) {
        return gnuais_batch_create(&s.b, s.device, s.n, taps, n_taps, pllinc, max_len, frame_capacity_per_device);
    });
    if (rc != GNUAIS_OK) {
        const std::string keep = g_node_err;
        gnuais_node_destroy(nd);
        return node_fail(rc, keep);
    }
    // not fatal, but worth knowing before a scaling curve is read: a shard whose host thread stayed where the OS put it
    // (no NUMA node in sysfs, no CPU of that node in this process's affinity mask, GNUAIS_NODE_PIN=0)
    for (size_t g = 0; g < nd->shards.size(); ++g) {
        const Shard *s = nd->shards[g];
        if (s->pinned_cpus == 0 && nd->shards.size() > 1) {
            char line[200];
            snprintf(line, sizeof line, "shard %zu (device %d, pci %s): host thread not pinned (numa_node %d)\n", g, s->device,
                     s->pci[0] ? s->pci : "?", s->numa_node);
            nd->warnings += line;
        }
    }
    *out = nd;
    return GNUAIS_OK;
}

const char *gnuais_node_warnings(const gnuais_node *nd) { return nd ? nd->warnings.c_str() : ""; }

int gnuais_node_n_devices(const gnuais_node *nd) { return nd ? (int) nd->shards.size() : 0; }
int gnuais_node_n_channels(const gnuais_node *nd) { return nd ? nd->N : 0; }

int gnuais_node_shard(const gnuais_node *nd, int i, int *device, int *first_channel, int *n_channels,
                      gnuais_batch **batch)
{
    if (!nd || i < 0 || i >= (int) nd->shards.size()) return node_fail(GNUAIS_E_ARG, "node_shard: index");
    const Shard *s = nd->shards[i];
    if (device) *device = s->device;
    if (first_channel) *first_channel = s->first;
    if (n_channels) *n_channels = s->n;
    if (batch) *batch = s->b;
    return GNUAIS_OK;
}

int gnuais_node_reset(gnuais_node *nd)
{
    if (!nd) return node_fail(GNUAIS_E_ARG, "node_reset: NULL");
    return run_all(nd, [](Shard &s) { return gnuais_batch_reset(s.b); });
}

int gnuais_node_run(gnuais_node *nd, const int16_t *const *d_samples, int len, void *const *streams)
{
    if (!nd || !d_samples) return node_fail(GNUAIS_E_ARG, "node_run: NULL argument");
    if (len <= 0 || len > nd->max_len) return node_fail(GNUAIS_E_ARG, "node_run: len out of range");
    std::vector<Shard *> &sh = nd->shards;
    return run_all(nd, [&](Shard &s) {
        const size_t i = (size_t) (std::find(sh.begin(), sh.end(), &s) - sh.begin());
        const double t0 = wall_ms();
        const int rc = gnuais_batch_run(s.b, d_samples[i], len, streams ? streams[i] : nullptr);
        if (s.m_calls++ == 0) s.m_first = t0;
        s.m_submit_ms += wall_ms() - t0;
        return rc;
    });
}

int gnuais_node_run_host(gnuais_node *nd, const int16_t *h_samples, int len)
{
    if (!nd || !h_samples) return node_fail(GNUAIS_E_ARG, "node_run_host: NULL argument");
    if (len <= 0 || len > nd->max_len) return node_fail(GNUAIS_E_ARG, "node_run_host: len out of range");
    const int N = nd->N, max_len = nd->max_len;
    return run_all(nd, [=](Shard &s) -> int {
        const double t0 = wall_ms();
        if (s.m_calls++ == 0) s.m_first = t0;
        struct Stop { Shard &s; double t0; ~Stop() { s.m_submit_ms += wall_ms() - t0; } } stop{s, t0};
        if (hipSetDevice(s.device) != hipSuccess) return node_fail(GNUAIS_E_HIP, "node_run_host: hipSetDevice");
        if (!s.s_in) {
            if (hipStreamCreateWithFlags(&s.s_in, hipStreamNonBlocking) != hipSuccess)
                return node_fail(GNUAIS_E_HIP, "node_run_host: stream");
            for (int q = 0; q < 2; ++q) {
                if (hipMalloc((void **) &s.d_in[q], sizeof(int16_t) * (size_t) max_len * (size_t) s.n) != hipSuccess ||
                    hipEventCreateWithFlags(&s.e_free[q], hipEventDisableTiming) != hipSuccess)
                    return node_fail(GNUAIS_E_HIP, "node_run_host: staging allocation");
            }
        }
        const int q = (int) (s.host_calls & 1);
        if (s.host_calls >= 2 && hipEventSynchronize(s.e_free[q]) != hipSuccess)
            return node_fail(GNUAIS_E_HIP, "node_run_host: wait for the staging slab");
        // the shard's columns of the interleaved buffer: a strided 2-D copy, rows of n channels out of N
        if (hipMemcpy2DAsync(s.d_in[q], sizeof(int16_t) * (size_t) s.n, h_samples + s.first,
                             sizeof(int16_t) * (size_t) N, sizeof(int16_t) * (size_t) s.n, (size_t) len,
                             hipMemcpyHostToDevice, s.s_in) != hipSuccess)
            return node_fail(GNUAIS_E_HIP, "node_run_host: host -> device copy");
        const int rc = gnuais_batch_run(s.b, s.d_in[q], len, s.s_in);
        (void) hipEventRecord(s.e_free[q], s.s_in);
        s.host_calls++;
        // the caller's buffer is borrowed for the call only (src/ais.c:216 reuses it): the copy out of it must be done
        if (hipStreamSynchronize(s.s_in) != hipSuccess && rc == GNUAIS_OK)
            return node_fail(GNUAIS_E_HIP, "node_run_host: copy");
        return rc;
    });
}

int gnuais_node_sync(gnuais_node *nd)
{
    if (!nd) return node_fail(GNUAIS_E_ARG, "node_sync: NULL");
    return run_all(nd, [](Shard &s) {
        const int rc = gnuais_batch_sync(s.b);
        s.m_last_sync = wall_ms();
        return rc;
    });
}

// Per-shard bookkeeping for whoever times a node (bench.py --gpus N): gnuais_node_mark() starts a measurement,
// gnuais_node_shard_stats() reads one shard's after a gnuais_node_sync().
int gnuais_node_mark(gnuais_node *nd)
{
    if (!nd) return node_fail(GNUAIS_E_ARG, "node_mark: NULL");
    return run_all(nd, [](Shard &s) {
        s.m_calls = 0;
        s.m_submit_ms = s.m_first = s.m_last_sync = 0.0;
        return GNUAIS_OK;
    });
}

int gnuais_node_shard_stats(const gnuais_node *nd, int i, gnuais_node_shard_stat *out)
{
    if (!nd || !out || i < 0 || i >= (int) nd->shards.size()) return node_fail(GNUAIS_E_ARG, "node_shard_stats: argument");
    const Shard *s = nd->shards[i];
    memset(out, 0, sizeof *out);
    out->device = s->device;
    out->first_channel = s->first;
    out->n_channels = s->n;
    out->numa_node = s->numa_node;
    out->pinned_cpus = s->pinned_cpus;
    out->calls = (long long) s->m_calls;
    out->submit_ms = s->m_submit_ms;
    out->busy_ms = (s->m_calls && s->m_last_sync > s->m_first) ? s->m_last_sync - s->m_first : 0.0;
    snprintf(out->pci, sizeof out->pci, "%s", s->pci);
    return GNUAIS_OK;
}

int gnuais_node_pending_frames(gnuais_node *nd, int *n_out)
{
    if (!nd || !n_out) return node_fail(GNUAIS_E_ARG, "node_pending_frames: argument");
    std::vector<int> n(nd->shards.size(), 0);
    std::vector<Shard *> &sh = nd->shards;
    const int rc = run_all(nd, [&](Shard &s) {
        const size_t i = (size_t) (std::find(sh.begin(), sh.end(), &s) - sh.begin());
        return gnuais_batch_pending_frames(s.b, &n[i]);
    });
    long long t = 0;
    for (int v : n) t += v;
    *n_out = (int) std::min<long long>(t, 0x7fffffff);
    return rc;
}

int gnuais_node_drain_frames(gnuais_node *nd, gnuais_frame *h_out, int max, int *n_out)
{
    if (!nd || !h_out || !n_out || max < 0) return node_fail(GNUAIS_E_ARG, "node_drain_frames: argument");
    *n_out = 0;
    int total = 0;
    if (int rc = gnuais_node_pending_frames(nd, &total)) return rc;
    if (total > max) return node_fail(GNUAIS_E_ARG, "node_drain_frames: frame buffer too small");
    // every device drains into its own part of the output, concurrently; a shard's records come in the reference's
    // order (channel, then time) with LOCAL channel numbers, so the parts only have to be laid end to end -- shards
    // own ascending channel blocks -- and renumbered
    std::vector<Shard *> &sh = nd->shards;
    std::vector<int> cnt(sh.size(), 0), off(sh.size(), 0);
    {
        std::vector<int> pend(sh.size(), 0);
        const int rc = run_all(nd, [&](Shard &s) {
            const size_t i = (size_t) (std::find(sh.begin(), sh.end(), &s) - sh.begin());
            return gnuais_batch_pending_frames(s.b, &pend[i]);
        });
        if (rc) return rc;
        int o = 0;
        for (size_t i = 0; i < sh.size(); ++i) { off[i] = o; o += pend[i]; }
        if (o > max) return node_fail(GNUAIS_E_ARG, "node_drain_frames: frame buffer too small");
        cnt = pend;
    }
    const int rc = run_all(nd, [&](Shard &s) {
        const size_t i = (size_t) (std::find(sh.begin(), sh.end(), &s) - sh.begin());
        int got = 0;
        const int r = gnuais_batch_drain_frames(s.b, h_out + off[i], cnt[i], &got);
        for (int k = 0; k < got; ++k) h_out[off[i] + k].channel += (uint32_t) s.first;
        cnt[i] = got;
        return r;
    });
    // close gaps if a device delivered fewer than it had announced (it cannot deliver more)
    int w = 0;
    for (size_t i = 0; i < sh.size(); ++i) {
        if (off[i] != w && cnt[i] > 0) memmove(h_out + w, h_out + off[i], sizeof(gnuais_frame) * (size_t) cnt[i]);
        w += cnt[i];
    }
    *n_out = w;
    return rc;              // GNUAIS_E_OVERFLOW of a device is reported with what was drained
}

// Streamed sentences of the whole node: gnuais_batch_stream_nmea() on every shard, each from its own thread.  Shard g's
// channels all lie before shard g+1's and a sentence does not name its channel (protodec.c:857-859: always 'A'), so the
// shards' texts written out in shard order are the node's sentences in the reference's order for that call.
int gnuais_node_stream_nmea(gnuais_node *nd, const char **texts, size_t *lens, int *n_sentences, int *n_frames)
{
    if (!nd || !texts || !lens) return node_fail(GNUAIS_E_ARG, "node_stream_nmea: NULL argument");
    std::vector<Shard *> &sh = nd->shards;
    std::vector<int> ns(sh.size(), 0), nf(sh.size(), -1);
    for (size_t i = 0; i < sh.size(); ++i) { texts[i] = nullptr; lens[i] = 0; }    // a shard that fails leaves nothing stale
    const int rc = run_all(nd, [&](Shard &s) {
        const size_t i = (size_t) (std::find(sh.begin(), sh.end(), &s) - sh.begin());
        return gnuais_batch_stream_nmea(s.b, &texts[i], &lens[i], &ns[i], &nf[i]);
    });
    int sent = 0, frames = 0;
    bool filling = false;
    for (size_t i = 0; i < sh.size(); ++i) {
        sent += ns[i];
        if (nf[i] < 0) filling = true; else frames += nf[i];
    }
    if (n_sentences) *n_sentences = sent;
    if (n_frames) *n_frames = filling ? -1 : frames;        // every shard fills and drains in the same call
    return rc;
}

int gnuais_node_discard_frames(gnuais_node *nd)
{
    if (!nd) return node_fail(GNUAIS_E_ARG, "node_discard_frames: NULL");
    return run_all(nd, [](Shard &s) { return gnuais_batch_discard_frames(s.b, nullptr); });
}

int gnuais_node_counters(gnuais_node *nd, gnuais_counters *h_out)
{
    if (!nd || !h_out) return node_fail(GNUAIS_E_ARG, "node_counters: argument");
    return run_all(nd, [&](Shard &s) { return gnuais_batch_counters(s.b, h_out + s.first); });
}

int gnuais_node_total_received(gnuais_node *nd, long long *total)
{
    if (!nd || !total) return node_fail(GNUAIS_E_ARG, "node_total_received: argument");
    std::vector<Shard *> &sh = nd->shards;
    std::vector<long long> t(sh.size(), 0);
    const int rc = run_all(nd, [&](Shard &s) {
        const size_t i = (size_t) (std::find(sh.begin(), sh.end(), &s) - sh.begin());
        return gnuais_batch_total_received(s.b, &t[i]);
    });
    *total = 0;
    for (long long v : t) *total += v;
    return rc;
}

int gnuais_node_maxval(gnuais_node *nd, int16_t *h_out)
{
    if (!nd || !h_out) return node_fail(GNUAIS_E_ARG, "node_maxval: argument");
    return run_all(nd, [&](Shard &s) { return gnuais_batch_maxval(s.b, h_out + s.first); });
}

int gnuais_node_pll_state(gnuais_node *nd, gnuais_pll_state *h_out)
{
    if (!nd || !h_out) return node_fail(GNUAIS_E_ARG, "node_pll_state: argument");
    return run_all(nd, [&](Shard &s) { return gnuais_batch_pll_state(s.b, h_out + s.first); });
}

int gnuais_node_set_option(gnuais_node *nd, const char *name, int value)
{
    if (!nd || !name) return node_fail(GNUAIS_E_ARG, "node_set_option: argument");
    return run_all(nd, [&](Shard &s) { return gnuais_batch_set_option(s.b, name, value); });
}

int gnuais_node_autotune(gnuais_node *nd, const int16_t *const *d_samples, int len, void *const *streams,
                         float *best_ms_max)
{
    if (!nd || !d_samples) return node_fail(GNUAIS_E_ARG, "node_autotune: argument");
    std::vector<Shard *> &sh = nd->shards;
    std::vector<float> ms(sh.size(), 0.0f);
    // one device after the other: the measurement of one batch must not see another batch's host thread at work
    int rc = GNUAIS_OK;
    for (size_t i = 0; i < sh.size() && rc == GNUAIS_OK; ++i) {
        Shard *s = sh[i];
        s->w.submit([&, s, i] { return gnuais_batch_autotune(s->b, d_samples[i], len, streams ? streams[i] : nullptr, &ms[i]); });
        rc = s->w.wait();
        if (rc != GNUAIS_OK) g_node_err = s->w.err;
    }
    if (best_ms_max) *best_ms_max = *std::max_element(ms.begin(), ms.end());
    return rc;
}

} // extern "C"
