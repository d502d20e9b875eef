"""Fingerprint of a kernel's machine code: sha256 over the instruction encodings of every gfx950 kernel in a hipcc object file whose
(demangled) name matches a regex.  profiles/pmc_*.json store it next to the HBM traffic they measured, and bench.py reports that
traffic only when the kernels it runs still hash the same -- a counter figure collected for another kernel body is not this run's.

    python tools/kernel_hash.py zpc_amd/lib/obj/mpm_slotted.o 'g2p2g_slot_kernel<8, 1, false>|slot_rehome|slot_commit'
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_hashes(obj, regex):
    """{demangled kernel name: first 16 hex digits of sha256(encodings)}; {} if the tools or the object are missing"""
    if not (os.path.exists(obj) and os.path.exists(os.path.join(LLVM, "clang-offload-bundler"))):
        return {}
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "p.fat"), os.path.join(t, "p.co")
        try:
            subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   "--input=" + fat, "--output=" + co, "--unbundle"])
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "-C", co], stdout=subprocess.PIPE, check=True).stdout.decode()
        except Exception:
            return {}
    out, name, h = {}, None, None
    pat = re.compile(regex)
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
        if m:
            if name is not None:
                out[name] = h.hexdigest()[:16]
            name = m.group(1) if pat.search(m.group(1)) else None
            h = hashlib.sha256()
            continue
        if name is not None and "//" in ln:
            enc = ln.split("//", 1)[1].split(":", 1)[-1].strip().split(" <")[0]   # "ADDR: W0 W1 ..." -> the encoding words
            h.update(enc.encode())
    if name is not None:
        out[name] = h.hexdigest()[:16]
    return out


def combined(obj, regex):
    """one hash over all matching kernels (sorted by name), or None"""
    d = kernel_hashes(obj, regex)
    if not d:
        return None
    h = hashlib.sha256()
    for k in sorted(d):
        h.update((k + ":" + d[k] + ";").encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    d = kernel_hashes(sys.argv[1], sys.argv[2])
    for k in sorted(d):
        print(d[k], k[:140])
    print("combined", combined(sys.argv[1], sys.argv[2]))
