// host_util.h — small host-side helpers of libhalart.so: error channel, RAII device buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace rt {

// HalaRendererError (src/error.rs:5-22): message of the last failed call on this thread.
void set_last_error(const std::string& msg);
const char* get_last_error();

#define RT_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      rt::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
      return HALA_ERR;                                                                       \
    }                                                                                        \
  } while (0)

#define RT_FAIL(msg)                 \
  do {                               \
    rt::set_last_error(msg);         \
    return HALA_ERR;                 \
  } while (0)

template <class T>
struct DeviceArray {
  T* ptr = nullptr;
  size_t count = 0;
  DeviceArray() = default;
  DeviceArray(const DeviceArray&) = delete;
  DeviceArray& operator=(const DeviceArray&) = delete;
  ~DeviceArray() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
  // returns hipSuccess or the failing code; contents are NOT initialised
  hipError_t resize(size_t n) {
    if (n == count && ptr) return hipSuccess;
    release();
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), (n ? n : 1) * sizeof(T));
    if (e == hipSuccess) count = n; else ptr = nullptr;
    return e;
  }
  hipError_t upload(const T* src, size_t n, hipStream_t s) {
    hipError_t e = resize(n);
    if (e != hipSuccess || n == 0) return e;
    return hipMemcpyAsync(ptr, src, n * sizeof(T), hipMemcpyHostToDevice, s);
  }
  size_t bytes() const { return count * sizeof(T); }
};

// ---- minimal JSON (for HalaRayTracingProgramDesc, src/raytracing_program.rs:25-55) -------------------------
struct JsonValue {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<JsonValue> items;                            // Array
  std::vector<std::pair<std::string, JsonValue>> members;  // Object
  const JsonValue* find(const std::string& key) const {
    for (const auto& m : members) if (m.first == key) return &m.second;
    return nullptr;
  }
};
// returns "" on success, else a message with the byte offset of the problem
std::string json_parse(const char* text, JsonValue* out);

}  // namespace rt

// roctx range around a host-side phase (rocprofv3 --marker-trace shows commit / update / refit / tile_allgather on the timeline)
namespace rt { void roctx_push(const char* name); void roctx_pop(); }  // dyn_api.h: no-ops when the profiler SDK is not installed
struct RtRange {
  explicit RtRange(const char* name) { rt::roctx_push(name); }
  ~RtRange() { rt::roctx_pop(); }
  RtRange(const RtRange&) = delete;
  RtRange& operator=(const RtRange&) = delete;
};
