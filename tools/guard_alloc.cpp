// guard_alloc.cpp -- test-only device allocator that puts UNMAPPED address space on both sides of every torch tensor, so a kernel that
// reads or writes outside a buffer it was handed faults on ANY box instead of only where the neighbouring addresses happen to be unmapped
// (round 3 shipped such a fault: it did not reproduce on the builder's boxes).  Loaded through torch.cuda.memory.CUDAPluggableAllocator by
// tests/conftest.py when DAE_GUARD_ALLOC is set (1 / "end": the buffer ENDS at the end of its mapping; "start": it begins at its start).
//   hipcc -shared -fPIC -O2 -o tools/libguard_alloc.so tools/guard_alloc.cpp
// Each allocation = its own HIP virtual-memory reservation [guard | mapped granules | guard].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

namespace {
struct Rec { void* base; size_t reserved, mapped; void* map_at; hipMemGenericAllocationHandle_t h; };
std::mutex g_mu;
std::unordered_map<void*, Rec> g_recs;
size_t g_gran = 0;
int g_at_start = -1;
size_t g_align = 256;

void die(const char* what, hipError_t e) {
    fprintf(stderr, "guard_alloc: %s failed: %s\n", what, hipGetErrorString(e));
    abort();
}
}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t) {
    if (size <= 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (!g_gran) {
        hipError_t e = hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum);
        if (e != hipSuccess || !g_gran) die("hipMemGetAllocationGranularity", e);
        const char* m = getenv("DAE_GUARD_ALLOC");
        g_at_start = (m && !strcmp(m, "start")) ? 1 : 0;
        if (const char* a = getenv("DAE_GUARD_ALIGN")) g_align = (size_t)atol(a);
        fprintf(stderr, "guard_alloc: granularity %zu, buffers at the %s of their mapping, alignment %zu\n", g_gran, g_at_start ? "start" : "end", g_align);
    }
    const size_t need = ((size_t)size + g_align - 1) / g_align * g_align;
    const size_t mapped = (need + g_gran - 1) / g_gran * g_gran;
    const size_t guard = g_gran > (2u << 20) ? g_gran : (2u << 20);
    Rec r;
    r.reserved = mapped + 2 * guard;
    r.mapped = mapped;
    hipError_t e = hipMemAddressReserve(&r.base, r.reserved, g_gran, nullptr, 0);
    if (e != hipSuccess) die("hipMemAddressReserve", e);
    e = hipMemCreate(&r.h, mapped, &prop, 0);
    if (e != hipSuccess) die("hipMemCreate", e);
    r.map_at = (char*)r.base + guard;
    e = hipMemMap(r.map_at, mapped, 0, r.h, 0);
    if (e != hipSuccess) die("hipMemMap", e);
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(r.map_at, mapped, &acc, 1);
    if (e != hipSuccess) die("hipMemSetAccess", e);
    void* p = g_at_start ? r.map_at : (void*)((char*)r.map_at + mapped - need);
    g_recs[p] = r;
    // DAE_GUARD_POISON=1: fresh memory is filled with 0xA5 (NaN-ish floats, huge indices), so code that relies on new device memory
    // being zero -- true on a fresh box, not guaranteed anywhere -- fails visibly
    static const bool poison = [] { const char* q = getenv("DAE_GUARD_POISON"); return q && *q == '1'; }();
    if (poison) {
        e = hipMemset(r.map_at, 0xA5, mapped);
        if (e != hipSuccess) die("hipMemset(poison)", e);
    }
    return p;
}

// Freed buffers are QUARANTINED, not unmapped: remapping a just-released address range inside one process returned stale data to
// torch's own copy kernels on the test box (first cut of this file), so a range stays mapped until the quarantine exceeds
// DAE_GUARD_QUARANTINE_GB (default 160), then the oldest ranges are released.  Out-of-bounds detection does not depend on it.
#include <deque>
namespace {
std::deque<Rec> g_quarantine;
size_t g_quarantine_bytes = 0;
}
extern "C" void guard_free(void* ptr, size_t, hipStream_t) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_recs.find(ptr);
    if (it == g_recs.end()) { fprintf(stderr, "guard_alloc: free of unknown pointer %p\n", ptr); return; }
    g_quarantine.push_back(it->second);
    g_quarantine_bytes += it->second.mapped;
    g_recs.erase(it);
    static size_t cap = 0;
    if (!cap) { const char* q = getenv("DAE_GUARD_QUARANTINE_GB"); cap = (size_t)(q ? atol(q) : 160) << 30; }
    if (g_quarantine_bytes > cap) {
        (void)hipDeviceSynchronize();
        while (g_quarantine_bytes > cap / 2 && !g_quarantine.empty()) {
            const Rec r = g_quarantine.front();
            g_quarantine.pop_front();
            g_quarantine_bytes -= r.mapped;
            (void)hipMemUnmap(r.map_at, r.mapped);
            (void)hipMemRelease(r.h);
            (void)hipMemAddressFree(r.base, r.reserved);
        }
    }
}
