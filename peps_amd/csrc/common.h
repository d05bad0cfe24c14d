// Shared helpers for the pepsgpu device library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
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

// Size-bucketed caching allocator: tensors of the walker batch recur with identical sizes every
// row absorption, so hipMalloc is hit only during the first pass.
class Arena {
 public:
  ~Arena() { release(); }
  void *alloc(size_t bytes) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~size_t(255);
    auto it = free_.find(bytes);
    if (it != free_.end() && !it->second.empty()) {
      void *p = it->second.back();
      it->second.pop_back();
      live_[p] = bytes;
      return p;
    }
    void *p = nullptr;
    PG_CHECK_HIP(hipMalloc(&p, bytes));
    total_ += bytes;
    live_[p] = bytes;
    return p;
  }
  void free(void *p) {
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) return;
    free_[it->second].push_back(p);
    live_.erase(it);
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
  std::map<void *, size_t> live_;
  size_t total_ = 0;
};

}  // namespace pepsgpu
