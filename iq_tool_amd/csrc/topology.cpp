// topology.cpp -- where a GPU sits in the host: PCI address and NUMA node of HIP device `ordinal`, read from sysfs WITHOUT a HIP
// call, and the calling thread bound to that node's CPUs.  The reference has no counterpart (it is one host process whose three
// stage threads never leave the CPU, src/pipeline.c:96-116); a process (bench.py) or a shard thread (iqgpu_run) that feeds a GPU
// through pinned buffers owns this: on a two-socket node half of the H2D / D2H streams would otherwise cross the socket link.
// It has to happen before the first GPU call -- the runtime's helper threads inherit the mask -- and before the pinned buffers
// are allocated, which is why nothing here may initialise HIP.
//
// Device order: the KFD topology (/sys/class/kfd/kfd/topology/nodes/<n>/properties) lists the agents in the order the ROCm
// runtime enumerates them; nodes with simd_count > 0 are GPUs.  ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES (or
// CUDA_VISIBLE_DEVICES), each a list of indices, select and reorder them the way the runtime does; a list with anything but
// indices (UUIDs) is not interpreted: the lookup then fails and nothing is bound.  bench.py checks the answer against
// hipDeviceGetPCIBusId once the runtime is up.
#include <dirent.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>

#include "chain.hpp"

namespace {

struct GpuNode { int kfd_node; unsigned domain, location; };

// "/sys" -- or the root of a stand-in tree (iqgpu_debug_set("sysfs_root", path): tests/test_host_logic.py builds one to exercise the
// parsing and the binding on a machine without a GPU)
std::string sysfs_root()
{
    const std::string r = debug_value("sysfs_root");
    return r.empty() ? std::string("/sys") : r;
}

bool read_properties(int node, unsigned long long *simd, unsigned long long *loc, unsigned long long *dom)
{
    char path[512];
    snprintf(path, sizeof(path), "%s/class/kfd/kfd/topology/nodes/%d/properties", sysfs_root().c_str(), node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char key[64]; unsigned long long v;
    *simd = 0; *loc = 0; *dom = 0;
    while (fscanf(f, "%63s %llu", key, &v) == 2) {
        if (!strcmp(key, "simd_count")) *simd = v;
        else if (!strcmp(key, "location_id")) *loc = v;
        else if (!strcmp(key, "domain")) *dom = v;
    }
    fclose(f);
    return true;
}

// applies one *_VISIBLE_DEVICES list of indices; false when the variable holds anything else
bool apply_visible(const char *env, std::vector<GpuNode> &g)
{
    const char *v = getenv(env);
    if (!v) return true;
    std::vector<GpuNode> out;
    const char *p = v;
    while (*p) {
        while (*p == ' ' || *p == ',') ++p;
        if (!*p) break;
        char *end = nullptr;
        const long i = strtol(p, &end, 10);
        if (end == p || (*end && *end != ',' && *end != ' ')) return false;      // a UUID or junk
        if (i < 0 || (size_t)i >= g.size()) break;                               // the runtime stops at the first invalid index
        out.push_back(g[(size_t)i]);
        p = end;
    }
    g.swap(out);
    return true;
}

int lookup(int ordinal, GpuNode *out)
{
    std::vector<int> nodes;
    char dir[512];
    snprintf(dir, sizeof(dir), "%s/class/kfd/kfd/topology/nodes", sysfs_root().c_str());
    if (DIR *d = opendir(dir)) {
        while (struct dirent *e = readdir(d)) {
            char *end = nullptr;
            const long n = strtol(e->d_name, &end, 10);
            if (end != e->d_name && *end == 0) nodes.push_back((int)n);
        }
        closedir(d);
    } else {
        return fail(IQGPU_ENODEV, "no KFD topology in sysfs (%s)", dir);
    }
    std::sort(nodes.begin(), nodes.end());
    std::vector<GpuNode> gpus;
    for (int n : nodes) {
        unsigned long long simd, loc, dom;
        if (read_properties(n, &simd, &loc, &dom) && simd > 0) gpus.push_back(GpuNode{n, (unsigned)dom, (unsigned)loc});
    }
    if (!apply_visible("ROCR_VISIBLE_DEVICES", gpus)) return fail(IQGPU_EUNSUPPORTED, "ROCR_VISIBLE_DEVICES is not a list of indices: device order not derived");
    const char *hv = getenv("HIP_VISIBLE_DEVICES") ? "HIP_VISIBLE_DEVICES" : "CUDA_VISIBLE_DEVICES";
    if (!apply_visible(hv, gpus)) return fail(IQGPU_EUNSUPPORTED, "%s is not a list of indices: device order not derived", hv);
    if (ordinal < 0 || (size_t)ordinal >= gpus.size()) return fail(IQGPU_ENODEV, "device_ordinal %d out of range (%zu GPUs in the KFD topology)", ordinal, gpus.size());
    *out = gpus[(size_t)ordinal];
    return IQGPU_OK;
}

void bdf_of(const GpuNode &g, char *buf, size_t cap)
{
    snprintf(buf, cap, "%04x:%02x:%02x.%x", g.domain & 0xffffu, (g.location >> 8) & 0xffu, (g.location >> 3) & 0x1fu, g.location & 7u);
}

bool read_line(const char *path, char *buf, size_t cap)
{
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const bool ok = fgets(buf, (int)cap, f) != nullptr;
    fclose(f);
    if (ok) { size_t n = strlen(buf); while (n && (buf[n - 1] == '\n' || buf[n - 1] == ' ')) buf[--n] = 0; }
    return ok;
}

// "0-63,128-191" -> cpu_set_t
bool parse_cpulist(const char *s, cpu_set_t *set)
{
    CPU_ZERO(set);
    int n_set = 0;
    const char *p = s;
    while (*p) {
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) return false;
        long b = a;
        p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); if (end == p + 1) return false; p = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, set); ++n_set; }
        if (*p == ',') ++p;
    }
    return n_set > 0;
}

} // namespace

extern "C" int iqgpu_device_numa_node(int ordinal, int *node, char *pci_bus_id, size_t cap)
{
    if (!node) return fail(IQGPU_EINVAL, "iqgpu_device_numa_node: NULL argument");
    *node = -1;
    if (pci_bus_id && cap) pci_bus_id[0] = 0;
    GpuNode g{};
    const int rc = lookup(ordinal, &g);
    if (rc) return rc;
    char bdf[32], path[512], line[64];
    bdf_of(g, bdf, sizeof(bdf));
    if (pci_bus_id && cap) snprintf(pci_bus_id, cap, "%s", bdf);
    snprintf(path, sizeof(path), "%s/bus/pci/devices/%s/numa_node", sysfs_root().c_str(), bdf);
    if (!read_line(path, line, sizeof(line))) return fail(IQGPU_ENODEV, "%s not readable", path);
    *node = atoi(line);
    return IQGPU_OK;
}

extern "C" int iqgpu_bind_thread_to_device(int ordinal, int *node)
{
    int nd = -1;
    char bdf[32];
    if (node) *node = -1;
    int rc = iqgpu_device_numa_node(ordinal, &nd, bdf, sizeof(bdf));
    if (rc) return rc;
    if (node) *node = nd;
    if (nd < 0) return IQGPU_OK;                          // a single-node host (or a VM that hides the topology): nothing to bind to
    char path[512], line[1024];
    snprintf(path, sizeof(path), "%s/bus/pci/devices/%s/local_cpulist", sysfs_root().c_str(), bdf);
    cpu_set_t local, allowed, both;
    if (!read_line(path, line, sizeof(line)) || !parse_cpulist(line, &local)) return fail(IQGPU_ENODEV, "%s not usable", path);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return fail(IQGPU_EINVAL, "sched_getaffinity failed");
    CPU_AND(&both, &local, &allowed);
    if (CPU_COUNT(&both) == 0) return fail(IQGPU_EINVAL, "none of the CPUs next to %s (%s) is in this thread's affinity mask", bdf, line);
    if (sched_setaffinity(0, sizeof(both), &both) != 0) return fail(IQGPU_EINVAL, "sched_setaffinity to %s failed", line);
    // ... and its memory: pages this thread touches first (staging buffers, what hipHostMalloc pins) come from that node when it
    // has room.  MPOL_PREFERRED = 1; best effort (a container may forbid the call: the CPU binding alone already gives first-touch)
#ifdef SYS_set_mempolicy
    if (nd < 1024) {
        unsigned long mask[16] = {0};
        mask[nd / (8 * sizeof(unsigned long))] |= 1ul << (nd % (8 * sizeof(unsigned long)));
        (void)syscall(SYS_set_mempolicy, 1, mask, (unsigned long)(sizeof(mask) * 8));
    }
#endif
    return IQGPU_OK;
}
