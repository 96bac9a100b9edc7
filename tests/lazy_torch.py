"""torch, torch.distributed and torch.multiprocessing as LAZY module proxies for the CPU tests and their test doubles.

`pytest -m gpu` collects every test module of this directory. A module-level `import torch` would load the ROCm runtime
bundled with the torch wheel (7.0.2: libamdhip64 / libhsa-runtime64 / librccl under torch/lib) into the process BEFORE
libetx_hip.so - which is built and linked against /opt/rocm (7.2) - and the dynamic loader would then bind the product to the
older runtime by soname. With these proxies torch is imported by the first test that actually uses it (the gloo tests, which are
not marked gpu); a `-m gpu` process never imports it (tests/test_gpu_runtime.py asserts that)."""
import importlib


class LazyModule:
    def __init__(self, name):
        self.__dict__["_name"] = name
        self.__dict__["_module"] = None

    def __getattr__(self, attribute):
        if attribute.startswith("_"):
            # pytest's collector probes every module-level object of a test module (__test__, _pytestfixturefunction, ...): none of that may import torch
            raise AttributeError(attribute)
        if self.__dict__["_module"] is None:
            self.__dict__["_module"] = importlib.import_module(self.__dict__["_name"])
        return getattr(self.__dict__["_module"], attribute)


torch = LazyModule("torch")
dist = LazyModule("torch.distributed")
mp = LazyModule("torch.multiprocessing")
