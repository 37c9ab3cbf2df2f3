"""Import shim that lets the reference's arithmetic leaf modules execute VERBATIM.

TEST INFRASTRUCTURE ONLY (used by oracle/make_golden.py and tests that run in the
build container).  It only works where /root/reference is mounted; on the GPU box
it raises ReferenceUnavailable and callers skip.

What is shimmed (containers / plumbing only, no arithmetic):
  * paramz 0.9.1 (requirements.txt:15, not vendored, not installable offline):
    Param (ndarray subclass), Parameterized (no-op link_parameter[s]), ObsAr,
    caching.Cache_this (identity decorator), transformations.Logexp (dummy),
    parameterized.ParametersChangedMeta (= type).
  * parent packages GPy, GPy.util, GPy.core, GPy.core.parameterization,
    GPy.inference(.latent_function_inference), GPy.kern(.src) and GPyOpt,
    GPyOpt.acquisitions, GPyOpt.core(.task), GPyOpt.models are registered as
    synthetic module objects whose __path__ points INTO /root/reference, so the
    real __init__.py files (which import paramz-heavy glue, plotting, pathos, cma
    ...) are skipped while every submodule below is loaded from the reference file.
  * pathos.multiprocessing.ProcessingPool -> serial stand-in (map = list(map)).

What runs verbatim from /root/reference (never copied into this repo):
  GPy/util/{linalg,diag,config}.py, GPy/kern/src/{kern,stationary,rbf,se}.py,
  GPy/inference/latent_function_inference/{posterior,exact_gaussian_inference}.py,
  GPy/inference/mcmc/hmc.py, GPy/core/parameterization/priors.py,
  GPyOpt/acquisitions/base.py, GPyOpt/core/task/cost.py,
  maEI.py, maPI.py, uEI_noiseless.py, uPI.py, EI.py, PI.py, utility.py,
  parameter_distribution.py.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = os.environ.get("BOCF_REFERENCE", "/root/reference")


class ReferenceUnavailable(RuntimeError):
    pass


_installed = False


def _mod(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def install():
    """Register the shim modules.  Idempotent."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(os.path.join(REF, "GPy")):
        raise ReferenceUnavailable("reference tree not present at %s" % REF)
    sys.dont_write_bytecode = True  # /root/reference is read-only

    # ---- inert paramz ----------------------------------------------------
    class Param(np.ndarray):
        def __new__(cls, name, input_array, *a, **kw):
            obj = np.atleast_1d(np.asarray(input_array, dtype=float)).copy().view(cls)
            obj._name = name
            return obj

        def __array_finalize__(self, obj):
            self._name = getattr(obj, "_name", None)

        @property
        def values(self):
            return np.asarray(self)

    class Parameterized(object):
        def __init__(self, name=None, *a, **kw):
            self.name = name

        def link_parameter(self, *a, **kw):
            pass

        def link_parameters(self, *a, **kw):
            pass

        def unlink_parameter(self, *a, **kw):
            pass

    class ObsAr(np.ndarray):
        def __new__(cls, input_array, *a, **kw):
            return np.atleast_1d(np.asarray(input_array, dtype=float)).view(cls)

    def Cache_this(*a, **kw):
        def deco(f):
            return f
        return deco

    class Logexp(object):
        pass

    _mod("paramz", Param=Param, Parameterized=Parameterized, ObsAr=ObsAr)
    _mod("paramz.caching", Cache_this=Cache_this)
    _mod("paramz.transformations", Logexp=Logexp, __fixed__="fixed")
    _mod("paramz.parameterized", ParametersChangedMeta=type)
    _mod("paramz.domains", _REAL="real", _POSITIVE="positive", _NEGATIVE="negative", _BOUNDED="bounded")

    # ---- synthetic GPy parents ---------------------------------------------
    g = os.path.join(REF, "GPy")
    _mod("GPy", g)
    _mod("GPy.util", os.path.join(g, "util"))
    _mod("GPy.core", None, Param=Param, Parameterized=Parameterized)
    # __path__ set so that priors.py (Gamma prior of GPModel, gpmodel.py:67-68) loads from the reference file
    _mod("GPy.core.parameterization", os.path.join(g, "core", "parameterization"), Param=Param, Parameterized=Parameterized)
    _mod("GPy.core.parameterization.parameterized", None, Parameterized=Parameterized)
    _mod("GPy.core.parameterization.variational", None,
         VariationalPosterior=type("VariationalPosterior", (), {}))
    _mod("GPy.inference", os.path.join(g, "inference"))
    _mod("GPy.inference.latent_function_inference",
         os.path.join(g, "inference", "latent_function_inference"),
         LatentFunctionInference=type("LatentFunctionInference", (), {}))
    _mod("GPy.kern", os.path.join(g, "kern"))
    _mod("GPy.kern.src", os.path.join(g, "kern", "src"))
    _mod("GPy.kern.src.psi_comp", None, PSICOMP_RBF=lambda *a, **k: None,
         PSICOMP_RBF_GPU=lambda *a, **k: None, PSICOMP_GH=lambda *a, **k: None)
    _mod("GPy.kern.src.grid_kerns", None, GridRBF=type("GridRBF", (), {}))

    # GPy/util/config.py:28 uses ConfigParser.readfp (removed in 3.12; present on 3.10)
    import configparser
    if not hasattr(configparser.ConfigParser, "readfp"):
        configparser.ConfigParser.readfp = configparser.ConfigParser.read_file
    cfg = importlib.import_module("GPy.util.config")
    cfg.config.set("cython", "working", "False")
    importlib.import_module("GPy.util.diag")
    importlib.import_module("GPy.util.linalg")

    # ---- synthetic GPyOpt parents + pathos -----------------------------------
    o = os.path.join(REF, "GPyOpt")
    _mod("GPyOpt", o)
    _mod("GPyOpt.acquisitions", os.path.join(o, "acquisitions"))
    _mod("GPyOpt.core", os.path.join(o, "core"))
    _mod("GPyOpt.core.task", os.path.join(o, "core", "task"))
    _mod("GPyOpt.models", None, GPModel=type("GPModel", (), {}))

    class ProcessingPool(object):
        def __init__(self, *a, **kw):
            pass

        def map(self, f, xs):
            return [f(x) for x in xs]

    _mod("pathos", None)
    _mod("pathos.multiprocessing", None, ProcessingPool=ProcessingPool)
    _installed = True


def ref_module(name):
    """Import a dotted module of the reference packages (GPy.*, GPyOpt.*)."""
    install()
    return importlib.import_module(name)


def ref_toplevel(name):
    """Load a top-level reference script (maEI, uEI_noiseless, utility ...) by file
    path, without putting /root/reference on sys.path (it holds a stray `numpy`
    directory that would shadow the real package)."""
    install()
    key = "_bocf_ref_" + name
    if key in sys.modules:
        return sys.modules[key]
    spec = importlib.util.spec_from_file_location(key, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[key] = m
    spec.loader.exec_module(m)
    return m


def available():
    return os.path.isdir(os.path.join(REF, "GPy"))
