"""The C-ABI library loads on a CPU-only box and exports every symbol include/alfalfa_amd.h declares;
the device half refuses to run without a GPU instead of falling back."""
import os
import re

import pytest

import alfalfa_amd as aa
from alfalfa_amd import capi
from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "alfalfa_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aa_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    L = capi.lib()
    names = declared_symbols()
    assert len(names) >= 30
    bound = {n for n, _, _ in capi.SYMBOLS}
    for n in names:
        assert hasattr(L, n), "missing export " + n
        assert n in bound, "ctypes binding lacks " + n
    assert L.aa_abi_version() == 4


def test_struct_layout_matches_header():
    import ctypes as C
    assert capi.MB_INFO_DTYPE.itemsize == 80
    assert C.sizeof(capi.FrameHeader) == 16 + 8 + 48 + 16


def test_geometry():
    assert capi.raster_geometry(1920, 1080) == (1920, 1088)
    assert capi.raster_geometry(175, 143) == (176, 144)


@pytest.mark.skipif(capi.device_count() > 0, reason="a GPU is present")
def test_device_half_fails_loudly_without_gpu():
    with pytest.raises(aa.AlfalfaError) as e:
        aa.Context(0)
    assert e.value.kind == "NoDevice"


def test_host_cpus_is_what_the_process_can_use_and_prepare_sets_the_queue_count_once():
    """aa_host_cpus: hardware threads, affinity mask and cgroup CPU quota together (the round-4 GPU box shows 256 and grants 16);
    aa_runtime_prepare: GPU_MAX_HW_QUEUES unless the environment has it (the bindings call it when the library is loaded)."""
    L = capi.lib()
    n = L.aa_host_cpus()
    assert 1 <= n <= (os.cpu_count() or 1) and n <= len(os.sched_getaffinity(0))
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if a == "max" else int(a) / int(b)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / p if q > 0 else None
        except (OSError, ValueError):
            pass
    if quota:
        assert n <= max(1, round(quota))
    assert os.environ.get("GPU_MAX_HW_QUEUES")             # set when capi.lib() loaded the library (or by the caller before)
    assert L.aa_runtime_prepare() == 1                     # it is set now: a value that is there stands
