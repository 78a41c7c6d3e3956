"""diff-pruning_amd -- MI355X-native engine for the Taylor-importance hot path of Diff-Pruning.

Import by string (`importlib.import_module('diff-pruning_amd')`): the directory name mandated for this
package contains a hyphen.  Submodules:
  _lib        ctypes binding of libdp_hip.so (C-ABI in include/dp_hip.h); raises if the library is missing
  ops         tensor-level kernel wrappers
"""
__all__ = ['_lib', 'ops']
