// Shared helpers for the pepsgpu device library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace pepsgpu {

struct Error : public std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define PG_CHECK_HIP(expr)                                                                     \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      throw ::pepsgpu::Error(2, std::string("HIP error ") + hipGetErrorString(_e) + " at " +  \
                                    __FILE__ + ":" + std::to_string(__LINE__) + " in " #expr); \
    }                                                                                          \
  } while (0)

#define PG_REQUIRE(cond, code, msg)                       \
  do {                                                    \
    if (!(cond)) throw ::pepsgpu::Error((code), (msg));   \
  } while (0)

// Dynamic LDS above 64 KiB needs an explicit opt-in per kernel; errors are cleared so that a
// refusal surfaces at the launch itself, not as a stale hipGetLastError().
constexpr size_t JACOBI_LDS_MAX = 128 * 1024;
inline void allow_dynamic_lds(const void *func, size_t bytes) {
  if (bytes > 48 * 1024) {
    (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    (void)hipGetLastError();
  }
}

// Size-bucketed caching allocator: tensors of the walker batch recur with (nearly) identical sizes every row
// absorption, so hipMalloc is hit only during the first pass.  Sizes are rounded up to 1/8-octave buckets (<= 12.5 %
// slack) so that the shapes of shrunk bonds, which differ by a few states from row to row, reuse each other's blocks
// instead of each adding a fresh hipMalloc while blocks of neighbouring sizes sit idle.  When the device refuses an
// allocation the cached blocks are returned to it and the allocation is tried once more.
class Arena {
 public:
  ~Arena() { release(); }
  static size_t bucket(size_t bytes) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes <= 65536) return bytes;
    int top = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = size_t(1) << (top - 3);
    return (bytes + step - 1) & ~(step - 1);
  }
  void *alloc(size_t bytes) {
    bytes = bucket(bytes);
    auto it = free_.find(bytes);
    if (it != free_.end() && !it->second.empty()) {
      void *p = it->second.back();
      it->second.pop_back();
      live_[p] = {bytes, ++seq_};
      return p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
      (void)hipGetLastError();
      trim();                        // give the idle cached blocks back and retry once
      e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess)
      throw ::pepsgpu::Error(2, std::string("HIP error ") + hipGetErrorString(e) + " allocating " + std::to_string(bytes) +
                                    " bytes (" + std::to_string(total_) + " held by this context)");
    total_ += bytes;
    live_[p] = {bytes, ++seq_};
    return p;
  }
  // return every cached (not live) block to the device
  void trim() {
    for (auto &kv : free_)
      for (void *p : kv.second) { (void)hipFree(p); total_ -= kv.first; }
    free_.clear();
  }
  void free(void *p) {
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) return;
    free_[it->second.first].push_back(p);
    live_.erase(it);
  }
  // Exception safety of the temporaries of one engine operation (ArenaScope below): every block handed out after
  // mark() that is still live goes back to the cache.
  uint64_t mark() const { return seq_; }
  void rollback(uint64_t mark) {
    for (auto it = live_.begin(); it != live_.end();) {
      if (it->second.second > mark) { free_[it->second.first].push_back(it->first); it = live_.erase(it); }
      else ++it;
    }
  }
  void release() {
    for (auto &kv : free_)
      for (void *p : kv.second) (void)hipFree(p);
    free_.clear();
    for (auto &kv : live_) (void)hipFree(kv.first);
    live_.clear();
    total_ = 0;
  }
  size_t total_bytes() const { return total_; }

 private:
  std::map<size_t, std::vector<void *>> free_;
  std::map<void *, std::pair<size_t, uint64_t>> live_;   // block -> (bucket size, allocation sequence number)
  size_t total_ = 0;
  uint64_t seq_ = 0;
};

// Guards one engine operation whose allocations are either all returned to the caller on success or all garbage when
// it throws (a row absorption, an environment step, a replace-trace, the CG solve): on unwinding, what the operation
// allocated and still holds is returned to the arena instead of leaking for the life of the context.
class ArenaScope {
 public:
  explicit ArenaScope(Arena &a) : a_(a), mark_(a.mark()), exc_(std::uncaught_exceptions()) {}
  ~ArenaScope() { if (std::uncaught_exceptions() > exc_) a_.rollback(mark_); }
  ArenaScope(const ArenaScope &) = delete;
  ArenaScope &operator=(const ArenaScope &) = delete;

 private:
  Arena &a_;
  uint64_t mark_;
  int exc_;
};

}  // namespace pepsgpu
