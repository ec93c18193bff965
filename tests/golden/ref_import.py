"""Import the read-only reference (/root/reference) in THIS container with inert stubs.

Used only by make_golden.py (fixture generation) -- never by tests, smoke() or bench.py; the
reference does not exist on the GPU box.  No stub touches arithmetic: they replace absent
third-party modules (tensorflow, ml_collections, torchvision, ...) that the hot path does not
execute.
"""
import importlib
import importlib.machinery
import os
import sys
import types

REF = "/root/reference"


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True
    import torch
    import torch.utils.cpp_extension as cpp_ext

    _stub("overrides", overrides=lambda f: f)
    _stub("ml_collections", ConfigDict=ConfigDict)

    class _GFile:
        def __init__(self, *a, **k):
            raise RuntimeError("tf.io.gfile.GFile stub")

    gfile = types.SimpleNamespace(exists=os.path.exists, makedirs=lambda p: os.makedirs(p, exist_ok=True),
                                  GFile=_GFile, glob=lambda p: [])
    tf = _stub("tensorflow", io=types.SimpleNamespace(gfile=gfile))
    tf.config = types.SimpleNamespace(experimental=types.SimpleNamespace(set_visible_devices=lambda *a, **k: None))
    _stub("flowpp_models")
    _stub("torch._six", inf=float("inf"), container_abcs=importlib.import_module("collections.abc"),
          string_classes=(str,), int_classes=(int,))
    for n in ["torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.utils", "natsort",
              "tensorflow_datasets", "tensorflow_gan", "tensorflow_hub", "tensorflow.io", "tqdm_stub"]:
        if n not in sys.modules:
            _stub(n)
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision.utils"].make_grid = None
    sys.modules["torchvision.utils"].save_image = None
    sys.modules["natsort"].natsorted = sorted
    cpp_ext.load = lambda *a, **k: None  # the .cu ops are never built; CPU path uses upfirdn2d_native
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)  # flow_model.py:102 opens the wolf JSON by relative path
    return torch


def get_config(name="configs.vp.CIFAR10.indm_nll"):
    import torch
    cfg = importlib.import_module(name).get_config()
    cfg.device = torch.device("cpu")
    return cfg
