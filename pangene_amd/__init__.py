"""pangene_amd -- MI355X-native graph-construction path of lh3/pangene behind the pangene.h C surface.

The product is the C-ABI shared library `pangene_amd/lib/libpangene_amd.so` (host driver in C++ + HIP
kernels for gfx950) and the `pangene_amd/bin/pangene` command line built from `pangene_amd/csrc`.
This Python package is plumbing only: ctypes bindings (capi), synthetic PAF generators (synth) and
the torch.distributed exchange hook used when genomes are sharded over several GPUs (exchange).
"""
from . import capi, synth  # noqa: F401

__all__ = ["capi", "synth"]
