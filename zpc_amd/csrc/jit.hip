// jit.hip -- runtime compilation and module loading for the rocm backend: the hiprtc counterparts of the reference's NVRTC
// entry points (py_interop/cuda/Nvrtc.cpp:16-260): cuda_compile_program / cuda_load_module / cuda_unload_module /
// cuda_get_kernel / cuda_launch_kernel  ->  rocm_*.  `launch__device` (runtime.hip) takes the hipFunction_t these return.
// hiprtc is resolved lazily with dlopen so that the compute library itself never depends on the compiler being present.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"

namespace {

struct Hiprtc {
  void *lib = nullptr;
  int (*create)(void **prog, const char *src, const char *name, int numHeaders, const char **headers, const char **includeNames) = nullptr;
  int (*compile)(void *prog, int numOptions, const char **options) = nullptr;
  int (*logSize)(void *prog, size_t *) = nullptr;
  int (*log)(void *prog, char *) = nullptr;
  int (*codeSize)(void *prog, size_t *) = nullptr;
  int (*code)(void *prog, char *) = nullptr;
  int (*destroy)(void **prog) = nullptr;
  const char *(*errstr)(int) = nullptr;
  bool ok = false;
};
Hiprtc &hiprtc() {
  static Hiprtc h = [] {
    Hiprtc r;
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return r;
    auto sym = [&](const char *n) { return dlsym(r.lib, n); };
    r.create = (decltype(r.create))sym("hiprtcCreateProgram");
    r.compile = (decltype(r.compile))sym("hiprtcCompileProgram");
    r.logSize = (decltype(r.logSize))sym("hiprtcGetProgramLogSize");
    r.log = (decltype(r.log))sym("hiprtcGetProgramLog");
    r.codeSize = (decltype(r.codeSize))sym("hiprtcGetCodeSize");
    r.code = (decltype(r.code))sym("hiprtcGetCode");
    r.destroy = (decltype(r.destroy))sym("hiprtcDestroyProgram");
    r.errstr = (decltype(r.errstr))sym("hiprtcGetErrorString");
    r.ok = r.create && r.compile && r.logSize && r.log && r.codeSize && r.code && r.destroy;
    return r;
  }();
  return h;
}
bool check_rtc(int res, const char *what) {
  if (res == 0) return true;
  Hiprtc &h = hiprtc();
  fprintf(stderr, "Zpc-JIT error: %s failed: %s (%d)\n", what, h.errstr ? h.errstr(res) : "?", res);
  return false;
}

}  // namespace

extern "C" {

// cuda_compile_program (Nvrtc.cpp:29-146).  `arch`: gfx number (950 -> --offload-arch=gfx950); 0 = the current device.
// The output is always a code object (the PTX/CUBIN split of the reference has no rocm counterpart).  Returns 0 on success.
size_t rocm_compile_program(const char *src, int arch, const char *include_dir, bool debug, bool verbose, bool verify_fp,
                            bool fast_math, const char *output_path) {
  (void)verify_fp;
  Hiprtc &h = hiprtc();
  if (!h.ok) {
    fprintf(stderr, "Zpc-JIT error: libhiprtc.so could not be loaded\n");
    return (size_t)-1;
  }
  constexpr size_t max_path = 4096 + 16;
  if (include_dir && strlen(include_dir) > max_path) {
    fprintf(stderr, "Zpc-JIT error: include path too long (%zu)\n", strlen(include_dir));
    return (size_t)-1;
  }
  std::string arch_opt = "--offload-arch=";
  if (arch > 0) {
    arch_opt += "gfx" + std::to_string(arch);
  } else {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      fprintf(stderr, "Zpc-JIT error: no device to take the architecture from; pass arch explicitly (e.g. 950)\n");
      return (size_t)-1;
    }
    std::string name = prop.gcnArchName;  // "gfx950:sramecc+:xnack-"
    arch_opt += name.substr(0, name.find(':'));
  }
  std::string include_opt = std::string("-I") + (include_dir ? include_dir : ".");
  std::vector<const char *> opts;
  opts.push_back(arch_opt.c_str());
  opts.push_back(include_opt.c_str());
  opts.push_back("-std=c++17");
  opts.push_back("-DZS_ENABLE_ROCM=1");
  opts.push_back("-DPYZPC_EXEC_TAG=zs::rocm_c");
  opts.push_back("-DZPC_JIT_MODE");
  if (debug) {
    opts.push_back("-D_DEBUG");
    opts.push_back("-g");
  } else
    opts.push_back("-DNDEBUG");
  if (fast_math) opts.push_back("-ffast-math");

  void *prog = nullptr;
  int res = h.create(&prog, src, nullptr, 0, nullptr, nullptr);
  if (!check_rtc(res, "hiprtcCreateProgram")) return (size_t)res;
  res = h.compile(prog, (int)opts.size(), opts.data());
  if (res != 0 || verbose) {
    size_t n = 0;
    if (h.logSize(prog, &n) == 0 && n > 1) {
      std::vector<char> log(n + 1, 0);
      if (h.log(prog, log.data()) == 0) fprintf(res != 0 ? stderr : stdout, "%s\n", log.data());
    }
  }
  if (!check_rtc(res, "hiprtcCompileProgram")) {
    h.destroy(&prog);
    return (size_t)res;
  }
  size_t size = 0;
  res = h.codeSize(prog, &size);
  if (check_rtc(res, "hiprtcGetCodeSize")) {
    std::vector<char> out(size);
    res = h.code(prog, out.data());
    if (check_rtc(res, "hiprtcGetCode")) {
      FILE *f = fopen(output_path, "wb");
      if (f) {
        if (fwrite(out.data(), 1, size, f) != size) {
          fprintf(stderr, "Zpc-JIT error: failed to write output file '%s'\n", output_path);
          res = -1;
        }
        fclose(f);
      } else {
        fprintf(stderr, "Zpc-JIT error: failed to open output file '%s'\n", output_path);
        res = -1;
      }
    }
  }
  h.destroy(&prog);
  return (size_t)res;
}

// cuda_load_module (Nvrtc.cpp:153-236)
void *rocm_load_module(void *pol, const char *path) {
  zs_rocm_policy *p = (zs_rocm_policy *)pol;
  if (p && p->device >= 0) ZSR_CHECK(hipSetDevice(p->device));
  std::vector<char> input;
  FILE *f = fopen(path, "rb");
  if (!f) {
    fprintf(stderr, "Zpc-JIT error: failed to open input file '%s'\n", path);
    return nullptr;
  }
  fseek(f, 0, SEEK_END);
  const size_t len = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  input.resize(len);
  const bool okRead = fread(input.data(), 1, len, f) == len;
  fclose(f);
  if (!okRead) {
    fprintf(stderr, "Zpc-JIT error: failed to read input file '%s'\n", path);
    return nullptr;
  }
  hipModule_t module = nullptr;
  hipError_t e = hipModuleLoadData(&module, input.data());
  if (e != hipSuccess) {
    zsr::report_error(e, "hipModuleLoadData", __FILE__, __LINE__);
    fprintf(stderr, "Zpc-JIT error: loading code object '%s' failed\n", path);
    return nullptr;
  }
  return module;
}
void rocm_unload_module(void *pol, void *module) {  // Nvrtc.cpp:238-243
  (void)pol;
  ZSR_CHECK(hipModuleUnload((hipModule_t)module));
}
void *rocm_get_kernel(void *pol, void *module, const char *name) {  // Nvrtc.cpp:245-255
  (void)pol;
  hipFunction_t fn = nullptr;
  hipError_t e = hipModuleGetFunction(&fn, (hipModule_t)module, name);
  if (e != hipSuccess) {
    zsr::report_error(e, "hipModuleGetFunction", __FILE__, __LINE__);
    fprintf(stderr, "Zpc-JIT: failed to lookup kernel function %s in module\n", name);
    return nullptr;
  }
  return fn;
}
// cuda_launch_kernel (Nvrtc.cpp:257-271): block 256, grid ceil(dim / 256), on `stream`
size_t rocm_launch_kernel(void *context, void *kernel, size_t dim, void **args, void *stream) {
  (void)context;
  const unsigned block = 256, grid = (unsigned)((dim + block - 1) / block);
  hipError_t e = hipModuleLaunchKernel((hipFunction_t)kernel, grid, 1, 1, block, 1, 1, 0, (hipStream_t)stream, args, nullptr);
  if (e != hipSuccess) zsr::report_error(e, "hipModuleLaunchKernel", __FILE__, __LINE__);
  return (size_t)e;
}

}  // extern "C"
