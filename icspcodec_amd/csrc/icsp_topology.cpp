// icsp_topology.cpp — host side of multi-GPU placement (include/icsp_hip.h, "placement"): which NUMA node a device hangs off,
// which CPUs that node has, binding a host thread to them, and first-touch placement of a mapped range.
//
// The reference's only parallelism is a pool of host threads taking closed-GOP jobs (ICSP_thread.cpp:39-77,
// ICSP_Codec_Encoder_source.cpp:186-213); it never places anything.  Here a job's frames travel host -> device -> host by DMA
// from and into pinned file mappings, so on a two-socket 8-GPU node it matters that the pages and the threads that drive a
// device sit on the socket the device is attached to: otherwise half of the transfers cross the socket interconnect.
// Plain C++ and sysfs; no HIP, no libnuma (mbind is not needed: pages are placed by whoever touches them first, and the
// populating thread is bound to the right node).  Everything degrades to a no-op: unknown node (-1), a single node, a
// kernel without the files, a container that forbids sched_setaffinity.
#include <errno.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <string>
#include "icsp_hip.h"

namespace {
// Tests point this at a fake tree (ICSP_SYSFS_ROOT=/tmp/x makes /tmp/x/sys/... be read); read per call, host tools only.
std::string sys_path(const char* rel)
{
    const char* root = getenv("ICSP_SYSFS_ROOT");
    return std::string(root ? root : "") + rel;
}
bool read_line(const std::string& path, char* buf, size_t cap)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    const bool ok = fgets(buf, (int)cap, f) != nullptr;
    fclose(f);
    return ok;
}
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
} // namespace

extern "C" {

// "0-3,8,10-11" -> cpus[]; returns how many the list names (which may exceed cap: only cap are stored), -1 on a malformed list.
int icsp_parse_cpulist(const char* list, int* cpus, int cap)
{
    if (!list) return -1;
    int n = 0;
    const char* p = list;
    while (*p == ' ' || *p == '\t') p++;
    if (*p == 0 || *p == '\n') return 0;                       // an empty list (a memory-only node)
    for (;;) {
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0) return -1;
        long b = a;
        p = end;
        if (*p == '-') {
            p++;
            b = strtol(p, &end, 10);
            if (end == p || b < a) return -1;
            p = end;
        }
        if (b - a > 1 << 20) return -1;
        for (long c = a; c <= b; c++) { if (cpus && n < cap) cpus[n] = (int)c; n++; }
        while (*p == ' ' || *p == '\t') p++;
        if (*p == ',') { p++; continue; }
        if (*p == 0 || *p == '\n') return n;
        return -1;
    }
}

// NUMA node of the PCI function "dddd:bb:dd.f" (what hipDeviceGetPCIBusId prints), -1 when the kernel does not say.
int icsp_numa_node_of_pci(const char* bus_id)
{
    if (!bus_id || !*bus_id) return -1;
    std::string id(bus_id);
    for (auto& ch : id) if (ch >= 'A' && ch <= 'F') ch = (char)(ch - 'A' + 'a');      // sysfs names are lower case
    char buf[64];
    if (!read_line(sys_path("/sys/bus/pci/devices/") + id + "/numa_node", buf, sizeof(buf))) return -1;
    char* end = nullptr;
    const long v = strtol(buf, &end, 10);
    return (end == buf || v < 0) ? -1 : (int)v;
}

// CPUs of a NUMA node (cpus may be null to count); -1 when the node does not exist.
int icsp_numa_cpus(int node, int* cpus, int cap)
{
    if (node < 0) return -1;
    char buf[4096];
    if (!read_line(sys_path("/sys/devices/system/node/node") + std::to_string(node) + "/cpulist", buf, sizeof(buf))) return -1;
    return icsp_parse_cpulist(buf, cpus, cap);
}

// Number of NUMA nodes that have CPUs (1 when the kernel does not say).
int icsp_numa_nodes(void)
{
    int n = 0;
    for (int k = 0; k < 64; k++) if (icsp_numa_cpus(k, nullptr, 0) > 0) n++;
    return n > 0 ? n : 1;
}

// Binds the calling thread to the CPUs of `node`.  ICSP_OK also when there is nothing to do (node < 0, a single node) or the
// container forbids it (placement is an optimisation, never a requirement); *bound (may be null) says whether the mask changed.
int icsp_bind_thread_to_node(int node, int* bound)
{
    if (bound) *bound = 0;
    if (node < 0 || icsp_numa_nodes() < 2) return ICSP_OK;
    int tmp[4096];
    const int n = icsp_numa_cpus(node, tmp, 4096);
    if (n <= 0) return ICSP_OK;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int k = 0; k < n && k < 4096; k++) if (tmp[k] < CPU_SETSIZE) CPU_SET(tmp[k], &set);
    if (sched_setaffinity(0, sizeof(set), &set) != 0) return ICSP_OK;      // EPERM / EINVAL in a restricted container: stay put
    if (bound) *bound = 1;
    return ICSP_OK;
}

// First-touch placement: allocates the pages of [p, p + bytes) of a writable mapping from the calling thread, i.e. on the node
// the thread is bound to (MADV_POPULATE_WRITE; a touch per page on kernels before 5.14).  For ranges that are populated
// already (a MAP_POPULATE mapping, page-cache pages of an input file) this changes nothing.
int icsp_populate_here(void* p, size_t bytes)
{
    if (!p || !bytes) return ICSP_ERR_UNENOUGH_PARAM;
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    uint8_t* a = (uint8_t*)((uintptr_t)p & ~(uintptr_t)(page - 1));
    const size_t len = (size_t)((uint8_t*)p + bytes - a);
    if (madvise(a, len, MADV_POPULATE_WRITE) == 0) return ICSP_OK;
    if (errno != EINVAL) return ICSP_ERR_RANGE;                            // not a writable mapping, out of memory, ...
    for (size_t o = 0; o < len; o += page) { volatile uint8_t* q = a + o; *q = *q; }
    return ICSP_OK;
}

// Which device a chunk belongs to when a clip's chunks are dealt over `ndev` devices: round-robin in chunk order, so that the
// devices advance through the clip together (a chunk can be placed in the bitstream only when every chunk before it has
// reported its length) and a device's ranges of the mapped files are known in advance -- which is what lets them be placed
// on its node.
int icsp_chunk_device(int chunk, int ndev) { return ndev > 1 ? chunk % ndev : 0; }

} // extern "C"
