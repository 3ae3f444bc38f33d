"""ctypes loader for oracle/_build/liboracle.so (test infrastructure only)."""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
  """Compiles the C restatement with gcc (``oracle/Makefile``)."""
  if force or not os.path.exists(_SO):
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
  return _SO


def lib() -> ctypes.CDLL:
  global _lib
  if _lib is None:
    build()
    _lib = ctypes.CDLL(_SO)
  return _lib


def fptr(a: np.ndarray):
  assert a.dtype == np.float32 and a.flags.c_contiguous
  return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def iptr(a: np.ndarray):
  assert a.dtype == np.int64 and a.flags.c_contiguous
  return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def i64(x) -> ctypes.c_int64:
  return ctypes.c_int64(int(x))
