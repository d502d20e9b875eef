"""RocmExecutionPolicy -- host mirror of zs::CudaExecutionPolicy
(include/zensim/cuda/execution/ExecutionPolicy.cuh:345-911, py_interop/cuda/ExecutionPolicy.cpp:8-9).
Fluent setters return self, like the reference's `*this&`."""
from ._lib import lib


class RocmExecutionPolicy:
    def __init__(self):
        self._h = lib().policy__device()

    def __del__(self):
        try:
            if self._h:
                lib().del_policy__device(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # execution/ExecutionPolicy.hpp:110-126
    def sync(self, flag=True):
        lib().zs_rocm_policy_sync(self._h, int(bool(flag)))
        return self

    def profile(self, flag=True):
        lib().zs_rocm_policy_profile(self._h, int(bool(flag)))
        return self

    # cuda/execution/ExecutionPolicy.cuh:362-385
    def device(self, procid):
        lib().zs_rocm_policy_device(self._h, int(procid))
        return self

    def stream(self, streamid):
        lib().zs_rocm_policy_stream(self._h, int(streamid))
        return self

    def listen(self, incoming_proc, incoming_streamid):
        lib().zs_rocm_policy_listen(self._h, int(incoming_proc), int(incoming_streamid))
        return self

    def shmem(self, nbytes):
        lib().zs_rocm_policy_shmem(self._h, int(nbytes))
        return self

    def block(self, tpb):
        lib().zs_rocm_policy_block(self._h, int(tpb))
        return self

    def external_stream(self, hip_stream):
        """Run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
        lib().zs_rocm_policy_external_stream(self._h, hip_stream)
        return self

    def shouldSync(self):
        return bool(lib().zs_rocm_policy_should_sync(self._h))

    def getStream(self):
        return lib().zs_rocm_policy_get_stream(self._h)

    def syncCtx(self):
        lib().zs_rocm_policy_sync_ctx(self._h)

    def last_elapsed_ms(self):
        return float(lib().zs_rocm_policy_last_elapsed_ms(self._h))


def rocm_exec():
    """zs::cuda_exec() equivalent (cuda/execution/ExecutionPolicy.cuh:917-918)."""
    return RocmExecutionPolicy()
