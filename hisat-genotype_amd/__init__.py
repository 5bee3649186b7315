"""hgx: MI355X-native allele-compatibility scoring + EM abundance estimation.

Drop-in for the per-locus hot path of HISAT-genotype
(hisatgenotype_typing_core.typing / hisatgenotype_typing_common.single_abundance).
Compute runs in hand-written HIP kernels (csrc/) reached through the C-ABI in
include/hgx.h; there is no CPU fallback: a missing/unbuildable extension raises.
"""
__version__ = "0.1.0"

from .typing import single_abundance, type_locus, type_file, type_many, type_many_loci, typing, report_lines, typing_options  # noqa: E402,F401
from .locus import PackedLocus  # noqa: E402,F401
from .driver import genotyping_locus, run_panel  # noqa: E402,F401
from .results import build_tree, call_nuance_results, flatten, result_process  # noqa: E402,F401
