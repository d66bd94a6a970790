// topology.hip — host placement of a GPU's driver threads: which NUMA node a device hangs off, and binding the calling thread
// (and the pinned witness blocks it allocates afterwards: first touch) to that node's cores.
//
// Why it is on the path: with host-side witness generation a po2-20 SYN-A segment brings 0.94 GB over PCIe;
// at 40 segments/s per GPU that is ~40 GB/s per GPU out of host DRAM, 320 GB/s on an 8-GPU node.  A lane thread that runs on
// the other socket pulls every byte over the inter-socket fabric first.  Upstream leaves placement to the operator
// (`r0vm` is started per GPU by the caller: /root/reference/run-parallel.sh:15 pins nothing); here one process per GPU
// (bench.py --gpus N) and the session executor's lane threads (session.hip) bind themselves.
// Host code only: sysfs + sched_setaffinity + set_mempolicy (raw syscall: no libnuma in the image).
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <fstream>
#include <sstream>

#include <cctype>

#include "common.h"

using namespace zkh;

namespace {

std::string lower(std::string s) { for (auto& ch : s) ch = (char)tolower((unsigned char)ch); return s; }
bool read_line(const std::string& path, std::string* out) {
    std::ifstream f(path);
    if (!f) return false;
    std::getline(f, *out);
    return true;
}

}  // namespace

// "0-15,64-79" -> sorted CPU ids.  Rejects anything that is not a cpulist (so that a misread sysfs file never binds a thread to
// a garbage mask).
extern "C" const char* zkh_parse_cpulist(const char* text, int* cpus, size_t cap, size_t* n) {
    ZKH_REQUIRE(text && n, "parse_cpulist: bad argument");
    std::vector<int> out;
    std::string s(text);
    while (!s.empty() && (s.back() == '\n' || s.back() == ' ' || s.back() == '\r')) s.pop_back();
    std::stringstream ss(s);
    std::string part;
    while (std::getline(ss, part, ',')) {
        if (part.empty()) continue;
        char* end = nullptr;
        const long a = strtol(part.c_str(), &end, 10);
        long b = a;
        ZKH_REQUIRE(end != part.c_str() && a >= 0, "parse_cpulist: '%s' is not a cpulist", text);
        if (*end == '-') {
            const char* q = end + 1;
            b = strtol(q, &end, 10);
            ZKH_REQUIRE(end != q && b >= a, "parse_cpulist: '%s' is not a cpulist", text);
        }
        ZKH_REQUIRE(*end == 0 && b - a < 65536, "parse_cpulist: '%s' is not a cpulist", text);
        for (long c = a; c <= b; c++) out.push_back((int)c);
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    *n = out.size();
    if (cpus) for (size_t i = 0; i < out.size() && i < cap; i++) cpus[i] = out[i];
    return nullptr;
}

// NUMA node of PCI function `bdf` ("0000:c1:00.0") and that node's CPUs, read under `sysfs_root` ("/sys"; a test passes a
// fake tree).  *node = -1 when the kernel reports none (single-node hosts, most containers): then *n_cpus = 0.
extern "C" const char* zkh_pci_numa_cpus(const char* sysfs_root, const char* bdf, int* node, int* cpus, size_t cap, size_t* n_cpus) {
    ZKH_REQUIRE(sysfs_root && bdf && node && n_cpus, "pci_numa_cpus: bad argument");
    *node = -1; *n_cpus = 0;
    std::string line;
    const std::string root(sysfs_root);
    if (!read_line(root + "/bus/pci/devices/" + lower(bdf) + "/numa_node", &line)) return nullptr;      // no such device file: unknown
    char* end = nullptr;
    const long v = strtol(line.c_str(), &end, 10);
    if (end == line.c_str() || v < 0) return nullptr;
    if (!read_line(root + "/devices/system/node/node" + std::to_string(v) + "/cpulist", &line)) return nullptr;
    ZKH_TRY(zkh_parse_cpulist(line.c_str(), cpus, cap, n_cpus));
    if (*n_cpus) *node = (int)v;
    return nullptr;
}

extern "C" const char* zkh_device_numa_node(int device, int* node, char pci_bus_id[32]) {
    ZKH_REQUIRE(node, "device_numa_node: bad argument");
    char bdf[32] = {0};
    ZKH_HIP(hipDeviceGetPCIBusId(bdf, sizeof bdf, device));
    if (pci_bus_id) memcpy(pci_bus_id, bdf, 32);
    size_t n = 0;
    return zkh_pci_numa_cpus("/sys", bdf, node, nullptr, 0, &n);
}

// What tells one GPU from another across processes: the PCI bus id, the device's UUID (16 bytes as 32 hex digits; the same for
// every process whatever HIP_VISIBLE_DEVICES renumbering it runs under), its NUMA node and marketing name.  bench.py gathers these
// from every rank so that an N-GPU line can prove it ran on N distinct devices.
extern "C" const char* zkh_device_identity(int device, char pci_bus_id[32], char uuid_hex[40], int* numa_node, char name[64], int* visible_devices) {
    int count = 0;
    ZKH_HIP(hipGetDeviceCount(&count));
    if (visible_devices) *visible_devices = count;
    ZKH_REQUIRE(device >= 0 && device < count, "device_identity: device %d of %d visible", device, count);
    char bdf[32] = {0};
    ZKH_HIP(hipDeviceGetPCIBusId(bdf, sizeof bdf, device));
    if (pci_bus_id) memcpy(pci_bus_id, bdf, 32);
    if (uuid_hex) {
        hipUUID u;
        memset(&u, 0, sizeof u);
        memset(uuid_hex, 0, 40);
        if (hipDeviceGetUuid(&u, device) == hipSuccess) {
            // ROCm reports the 16 bytes as ASCII hex digits of the 64-bit unique id (what rocm-smi --showuniqueid prints): kept as
            // text when every byte is a printable hex digit, hex-encoded otherwise
            bool text = true;
            for (int i = 0; i < 16; i++) text = text && isxdigit((unsigned char)u.bytes[i]);
            if (text) memcpy(uuid_hex, u.bytes, 16);
            else for (int i = 0; i < 16; i++) snprintf(uuid_hex + 2 * i, 3, "%02x", (unsigned)(unsigned char)u.bytes[i]);
        } else (void)hipGetLastError();
    }
    if (name) {
        hipDeviceProp_t prop;
        memset(name, 0, 64);
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) snprintf(name, 64, "%s", prop.name);
        else (void)hipGetLastError();
    }
    if (numa_node) {
        *numa_node = -1;
        size_t n = 0;
        const char* e = zkh_pci_numa_cpus("/sys", bdf, numa_node, nullptr, 0, &n);
        if (e) zkh_free_error(e);
    }
    return nullptr;
}

// Bind the CALLING thread to the cores of `device`'s NUMA node, sliced when several devices share the node: with `share` > 1
// the thread gets slice `slot` of `share` equal slices of the node's CPU list (bench.py: the ranks of the GPUs on one socket
// split its cores; share = 1: the whole node).  Memory policy of the thread becomes "prefer that node" (pinned witness blocks
// and staging rings allocated afterwards land next to the GPU's root port).  Threads created afterwards inherit both.
// ZKH_AFFINITY=off disables it.  A host that reports no node for the device is left untouched: *node = -1, *n_cpus = 0.
extern "C" const char* zkh_bind_thread_to_device(int device, size_t slot, size_t share, int* node, size_t* n_cpus) {
    int nd = -1;
    size_t nc = 0;
    if (node) *node = -1;
    if (n_cpus) *n_cpus = 0;
    const char* env = getenv("ZKH_AFFINITY");
    if (env && !strcmp(env, "off")) return nullptr;
    char bdf[32] = {0};
    ZKH_HIP(hipDeviceGetPCIBusId(bdf, sizeof bdf, device));
    std::vector<int> cpus(4096);
    ZKH_TRY(zkh_pci_numa_cpus("/sys", bdf, &nd, cpus.data(), cpus.size(), &nc));
    if (nd < 0 || !nc) return nullptr;
    cpus.resize(std::min(nc, cpus.size()));
    // only CPUs this process may use (cgroup cpusets, an outer taskset): intersect with the current mask
    cpu_set_t cur;
    CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof cur, &cur) == 0) {
        std::vector<int> ok;
        for (int c : cpus) if (c < CPU_SETSIZE && CPU_ISSET(c, &cur)) ok.push_back(c);
        if (ok.empty()) return nullptr;                   // the node's cores are not ours to use: leave the thread alone
        cpus.swap(ok);
    }
    if (share > 1 && cpus.size() >= share) {
        const size_t per = cpus.size() / share, s = slot % share;
        cpus = std::vector<int>(cpus.begin() + s * per, cpus.begin() + (s + 1) * per);
    }
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) if (c < CPU_SETSIZE) CPU_SET(c, &set);
    ZKH_REQUIRE(sched_setaffinity(0, sizeof set, &set) == 0, "bind_thread_to_device: sched_setaffinity failed (errno %d)", errno);
#ifdef SYS_set_mempolicy
    if (nd < 1024) {
        unsigned long mask[16] = {0};
        mask[nd / (8 * sizeof(unsigned long))] |= 1ul << (nd % (8 * sizeof(unsigned long)));
        (void)syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, (unsigned long)(8 * sizeof mask));      // best effort: containers may forbid it
    }
#endif
    if (node) *node = nd;
    if (n_cpus) *n_cpus = cpus.size();
    return nullptr;
}
