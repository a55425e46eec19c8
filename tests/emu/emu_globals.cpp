// TEST INFRASTRUCTURE ONLY: storage for the HIP emulator's scheduler state (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
namespace emu {
BlockState* g_blk = nullptr;
Fiber* g_cur = nullptr;
}  // namespace emu
