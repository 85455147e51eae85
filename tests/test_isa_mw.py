"""CPU: the instruction order of the spread reduction's exchange, read off the SHIPPED library.

bbo_eig_mw.hpp publishes a piece as   data stores -> s_waitcnt 0 -> LDS arrival count -> flag store
and consumes one as                   flag loads (poll loop) -> data loads,
all with relaxed agent-scope atomics (`sc1` on gfx950) and no hardware fence.  The hardware keeps
that order for one wavefront; the compiler is held to it by __atomic_signal_fence in the source.
This test does not take the source's word for it: it disassembles bboptpy_amd/libbbopt_hip.so
(llvm-objdump from /opt/rocm, no GPU needed) and checks, in each of the three kernels,

  1. between the last agent-scope data store of a step and the `ds_add` of the arrival count
     stands an `s_waitcnt` that waits for vmcnt(0);
  2. the next agent-scope access after the `ds_add` is the flag STORE;
  3. no agent-scope LOAD stands between the flag store and the poll loop's load, and every other
     agent-scope load (the data) comes after the poll loop (its `s_sleep`);
  4. the wait is bounded by the constant-rate clock: an `s_memrealtime` inside the poll loop.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bboptpy_amd", "libbbopt_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = ("cma_tred_mw", "cma_tred_mw_chain", "cma_tred_mw512")


@pytest.fixture(scope="module")
def device_objects(tmp_path_factory):
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump under /opt/rocm")
    assert os.path.exists(LIB), "libbbopt_hip.so is not built"
    d = tmp_path_factory.mktemp("isa")
    so = os.path.join(d, "lib.so")
    shutil.copy(LIB, so)            # (--offloading writes the bundles NEXT to its input)
    subprocess.run([objdump, "--offloading", so], check=True, capture_output=True)
    objs = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "gfx950" in f]
    assert objs, "no gfx950 code objects in the library"
    return objdump, objs


def _disassemble(objdump, objs, kernel):
    for o in objs:
        syms = subprocess.run([objdump, "-t", o], check=True, capture_output=True, text=True).stdout
        m = re.search(r"\s(_ZN3bbo%d%sE\w*)\n" % (len(kernel), kernel), syms)
        if not m:
            continue
        out = subprocess.run([objdump, "-d", "--disassemble-symbols=" + m.group(1), o], check=True,
                             capture_output=True, text=True).stdout
        ins = []
        for line in out.splitlines():
            t = line.strip().split("//")[0].strip()
            if t and not t.endswith(":") and not t.startswith(("/", "Disassembly", "<")):
                ins.append(t)
        return ins
    raise AssertionError("kernel %s not found in the library" % kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_exchange_order_in_the_shipped_isa(device_objects, kernel):
    objdump, objs = device_objects
    ins = _disassemble(objdump, objs, kernel)
    assert len(ins) > 500, len(ins)
    agent = lambda t: " sc1" in t and " sc0" not in t          # relaxed agent scope (system: sc0 sc1)
    adds = [k for k, t in enumerate(ins) if t.startswith("ds_add_rtn_u32")]
    assert len(adds) == 1, "one arrival count expected, found %d" % len(adds)
    add = adds[0]
    # 1. data stores -> s_waitcnt vmcnt(0) -> ds_add
    stores_before = [k for k in range(add) if ins[k].startswith("global_store") and agent(ins[k])]
    assert stores_before, "no agent-scope data store in front of the arrival count"
    last_store = stores_before[-1]
    waits = [k for k in range(last_store + 1, add) if ins[k].startswith("s_waitcnt")
             and "vmcnt(0)" in ins[k]]
    assert waits, "no s_waitcnt vmcnt(0) between the last data store and the arrival count:\n" + \
        "\n".join(ins[last_store:add + 1])
    # 2. ds_add -> flag store
    after = [k for k in range(add + 1, len(ins)) if agent(ins[k])]
    assert after and ins[after[0]].startswith("global_store"), ins[after[0]] if after else None
    flag_store = after[0]
    # 3. flag store -> poll load ... s_sleep -> data loads
    sleeps = [k for k, t in enumerate(ins) if t.startswith("s_sleep")]
    assert len(sleeps) == 1, sleeps
    sleep = sleeps[0]
    assert flag_store < sleep
    loads = [k for k, t in enumerate(ins) if t.startswith("global_load") and agent(t)]
    polls = [k for k in loads if flag_store < k < sleep]
    assert len(polls) == 1, "exactly the poll's flag load between the flag store and s_sleep:\n" + \
        "\n".join(ins[k] for k in polls)
    assert any(ins[k].startswith("s_waitcnt") and "vmcnt(0)" in ins[k]
               for k in range(polls[0] + 1, sleep)), "the poll's result is not waited for"
    data_loads = [k for k in loads if k != polls[0]]
    assert data_loads and min(data_loads) > sleep, \
        "an exchange-buffer load in front of the poll loop's exit: %s" % ins[min(data_loads)]
    # (the stores of a step come before the loads of that step in address order as well)
    assert max(stores_before) < polls[0]
    # 4. the bounded wait reads the constant-rate clock inside the loop
    clk = [k for k, t in enumerate(ins) if t.startswith("s_memrealtime")]
    assert clk and polls[0] < clk[0] < min(data_loads), clk
