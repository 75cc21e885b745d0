// Dispatch trace: which kernels did the library launch?  Test instrumentation behind the C ABI (include/istgcn.h:
// istgcn_trace): off by default -- one relaxed atomic load per launch -- and, while on, every ISTGCN_LAUNCH notes the
// symbol of the kernel it is about to launch (hipKernelNameRefByPtr: the mangled name rocprofv3 shows) with a count.
// The parity tests switch it on around a call and assert that the kernel variant they mean to pin (the register-chained
// graph conv, the lean temporal conv, the 256-channel data gradient with dA, ...) is the one that ran; tools/kernel_coverage.py
// lists, for the whole GPU suite, every kernel of the library and the tests that launched it.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {
std::atomic<int> g_on{0};
std::mutex g_mu;
std::map<std::string, long long>& table() {
  static std::map<std::string, long long> t;
  return t;
}
}  // namespace

extern "C" int istgcn_trace_on() { return g_on.load(std::memory_order_relaxed); }

extern "C" void istgcn_trace_note(const void* kfn, const char* launcher) {
  const char* name = hipKernelNameRefByPtr(kfn, nullptr);
  (void)hipGetLastError();
  std::lock_guard<std::mutex> lk(g_mu);
  ++table()[(name && *name) ? std::string(name) : std::string("?") + (launcher ? launcher : "")];
}

// op 0: trace off, table cleared.  op 1: trace on, table cleared.  op 2: dump -- "count<TAB>symbol\n" per distinct kernel
// launched since the last clear into buf (NUL-terminated, truncated to cap); the trace stays as it is.
// Returns the number of bytes the full dump needs (without the NUL), or -1 for an unknown op.
extern "C" int istgcn_trace(int op, char* buf, int cap) {
  if (op == 0 || op == 1) {
    std::lock_guard<std::mutex> lk(g_mu);
    table().clear();
    g_on.store(op, std::memory_order_relaxed);
    return 0;
  }
  if (op != 2) return -1;
  std::string out;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const auto& kv : table()) out += std::to_string(kv.second) + "\t" + kv.first + "\n";
  }
  if (buf && cap > 0) {
    const size_t n = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (int)out.size();
}
