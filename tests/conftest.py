import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    # the oracle's CPU passes at full size dominate the GPU suite's wall time, and torch's default of one intra-op thread per core is the slowest setting
    # on the GPU boxes (128 cores: 5.1 s per oracle step against 0.20 s at 16 threads, bench.py's cpu_baseline sweep)
    import torch
    n = int(os.environ.get("GLOWTTS_TEST_THREADS", "16"))          # (0: torch's default)
    if n > 0:
        torch.set_num_threads(min(n, os.cpu_count() or 1))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "late(tier): run after the in-process oracle-parity tests - tier 1 = child-process / entry-point / loader "
                            "integration tests, tier 2 = process-group (RCCL / gloo) tests")


# Order of a run (the driver runs `pytest -m gpu -x`: whatever fails first hides everything behind it).  Round 4's run stopped at its third test - a
# process-group integration test that sorted in front of all 240 oracle comparisons.  Now: (0) every in-process oracle-parity test, in this file order -
# kernels first, then the decoder / encoder, then whole-model and benchmarked-size tests; (1) child-process, entry-point and loader tests; (2) tests that
# create a process group.  Within a tier the original order is kept.
_FILE_ORDER = ["test_abi", "test_oracle_golden", "test_gpu_mas", "test_gpu_conv", "test_gpu_prep", "test_gpu_wavenet_fused", "test_gpu_decoder",
               "test_gpu_decoder_fullwidth", "test_gpu_encoder", "test_gpu_optim", "test_gpu_model", "test_gpu_fullsize", "test_gpu_benchmarked_sizes"]


def pytest_collection_modifyitems(session, config, items):
    def key(entry):
        index, item = entry
        mark = item.get_closest_marker("late")
        tier = int(mark.args[0]) if (mark is not None and mark.args) else (1 if mark is not None else 0)
        stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        rank = _FILE_ORDER.index(stem) if stem in _FILE_ORDER else len(_FILE_ORDER)
        return (tier, rank, index)
    items[:] = [item for _, item in sorted(enumerate(items), key=key)]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_device_memory_between_tests():
    """Trainer / graph tests leave captured hipGraphs (with their private memory pools) behind in reference cycles: without a collection the next
    test's eager allocations fall through the caching allocator to hipMalloc / hipFree on every call (a 2-s test ran 100 s behind the Trainer tests)."""
    yield
    import gc
    import time
    t0 = time.time()
    gc.collect()
    t1 = time.time()
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.empty_cache()
    except Exception:
        pass
    if time.time() - t0 > 1.0 and os.environ.get("GLOWTTS_TEST_VERBOSE"):
        print(f"[teardown] gc.collect {t1 - t0:.1f} s, empty_cache {time.time() - t1:.1f} s")
