#!/bin/bash
# AddressSanitizer, then ThreadSanitizer, over the HOST units (front end, readers, locus tables) on the CPU build: the three .cpp units are rebuilt with
# -fsanitize=address, linked with the regular device objects into a scratch libhgx.so, the CPU front-end tests and a
# multi-threaded SAM / BAM parse run under it, and the regular library is put back.  (GPU sanitizers are not available on the pool.)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/hisat-genotype_amd/csrc
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()"
cp $C/libhgx.so /tmp/libhgx_normal.so
trap 'cp /tmp/libhgx_normal.so $C/libhgx.so' EXIT
for f in hgx_sam hgx_bam hgx_host; do
    g++ -pthread -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -I $R/include -I $C -c $C/$f.cpp -o /tmp/${f}_asan.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $C/libhgx.so /tmp/hgx_sam_asan.o /tmp/hgx_bam_asan.o /tmp/hgx_host_asan.o \
    $(ls $C/*.o | grep -v -e hgx_sam.o -e hgx_bam.o -e hgx_host.o) -lz -ldl
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1
cd $R
# (tests of the error paths throw C++ exceptions; under an LD_PRELOADed libasan in a Python process the __cxa_throw interceptor
#  trips over its own CHECK -- "REAL(__cxa_throw) != 0" -- so those are left to the regular build)
python3 -m pytest tests/test_bamio.py tests/test_host_pieces.py tests/test_frontend_golden.py -x -q \
    -k "not error and not malformed and not truncated and not missing and not refuses and not rejects and not quirk"
python3 - <<'PY'
import sys, os, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, locus as hl, bamio
loc = synth.make_hla_like_locus(n_alleles=1500, n_vars=900, seed=7)
pl = hl.PackedLocus.from_synth(loc)
sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 7), 120000, err_rate=0.003, seed=3)
d = tempfile.mkdtemp()
ps, pb = os.path.join(d, "x.sam"), os.path.join(d, "x.bam")
open(ps, "w").write(sam)
bamio.write_bam_native(pb, sam, [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
from hisatgenotype_amd import engine
engine.test_switch("bam_chain_min", "1000000")      # the record chain is walked in ranges
b1 = pl.parse_sam(sam, n_threads=8)
for b in (pl.parse_alignment_file(ps, None, n_threads=8), pl.parse_alignment_file(pb, None, n_threads=8),
          pl.parse_alignment_file(pb, [loc.ref_allele], n_threads=5)):
    assert (b.n_pairs, b.n_reads) == (b1.n_pairs, b1.n_reads)
    assert np.array_equal(np.asarray(b.pair_ref), np.asarray(b1.pair_ref)) and np.array_equal(np.asarray(b.masks), np.asarray(b1.masks))
print("ASan: SAM text, SAM file, sorted BAM and BAM with a region list give the same batch (%d pairs, %d pieces), no report" % (b1.n_pairs, len(b1.pieces)))
PY

# ---- ThreadSanitizer: the same multi-threaded parse (worker pool, shared novel-variant table, partitioned grouping and merges) ----
unset LD_PRELOAD ASAN_OPTIONS
for f in hgx_sam hgx_bam hgx_host; do
    g++ -pthread -O1 -g -fsanitize=thread -std=c++17 -fPIC -I $R/include -I $C -c $C/$f.cpp -o /tmp/${f}_tsan.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $C/libhgx.so /tmp/hgx_sam_tsan.o /tmp/hgx_bam_tsan.o /tmp/hgx_host_tsan.o \
    $(ls $C/*.o | grep -v -e hgx_sam.o -e hgx_bam.o -e hgx_host.o) -lz -ldl
LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=0" python3 - > /tmp/hgx_tsan.log 2>&1 <<'PY'
import sys, os, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, locus as hl, bamio
loc = synth.make_hla_like_locus(n_alleles=1500, n_vars=900, seed=7)
pl = hl.PackedLocus.from_synth(loc)
sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 7), 60000, err_rate=0.003, seed=3)
d = tempfile.mkdtemp()
ps, pb = os.path.join(d, "x.sam"), os.path.join(d, "x.bam")
open(ps, "w").write(sam)
bamio.write_bam_native(pb, sam, [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
os.environ["HGX_BAM_CHAIN_MIN"] = "1000000"
b1 = pl.parse_sam(sam, n_threads=8)
for b in (pl.parse_alignment_file(ps, None, n_threads=8), pl.parse_alignment_file(pb, None, n_threads=8)):
    assert (b.n_pairs, b.n_reads) == (b1.n_pairs, b1.n_reads) and np.array_equal(np.asarray(b.pair_ref), np.asarray(b1.pair_ref))
print("parsed", b1.n_pairs, "pairs three ways")
PY
echo "TSan: $(grep -c 'WARNING: ThreadSanitizer' /tmp/hgx_tsan.log || true) reports; $(tail -n 1 /tmp/hgx_tsan.log)"
