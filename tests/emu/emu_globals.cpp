// TEST INFRASTRUCTURE ONLY: storage for the HIP emulator's scheduler state (see hip/hip_runtime.h), and the allocation
// registry that buffer descriptors are checked against.
#include <hip/hip_runtime.h>

#include <map>
#include <string>
namespace emu {
BlockState* g_blk = nullptr;
Fiber* g_cur = nullptr;

static std::map<uintptr_t, uintptr_t> g_allocs;      // base -> end
static long g_violations = 0;
static std::string g_first_violation;

void descriptor_check(const void* base, uint32_t bytes) {
    if (g_allocs.empty() || !base || bytes == 0) return;
    const uintptr_t b = (uintptr_t)base;
    auto it = g_allocs.upper_bound(b);
    if (it == g_allocs.begin()) return;
    --it;
    if (b >= it->second) return;                      // not inside a registered allocation
    if (b + bytes > it->second) {
        if (g_violations++ == 0) {
            char msg[256];
            snprintf(msg, sizeof msg, "buffer descriptor [+%zu, +%zu) of a %zu-byte allocation: %zu bytes past its end",
                     (size_t)(b - it->first), (size_t)(b - it->first) + bytes, (size_t)(it->second - it->first),
                     (size_t)(b + bytes - it->second));
            g_first_violation = msg;
        }
    }
}
}  // namespace emu

extern "C" void ssn_emu_alloc_register(const void* base, long bytes) {
    if (base && bytes > 0) emu::g_allocs[(uintptr_t)base] = (uintptr_t)base + (uintptr_t)bytes;
}
extern "C" void ssn_emu_alloc_unregister(const void* base) { emu::g_allocs.erase((uintptr_t)base); }
// number of descriptor violations since the last call (reset); msg: the first one
extern "C" long ssn_emu_alloc_violations(char* msg, long msg_bytes) {
    const long n = emu::g_violations;
    if (msg && msg_bytes > 0) snprintf(msg, (size_t)msg_bytes, "%s", emu::g_first_violation.c_str());
    emu::g_violations = 0;
    emu::g_first_violation.clear();
    return n;
}
