#!/bin/bash
# Builds experimental variants of libraxtax_hip.so (rtx_hit_quad.hip with other constants / parts switched off) into
# gpurun_scratch/ -- run on the build box; the GPU box then times them with tools/quad_time.py.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_scratch; mkdir -p "$OUT"
python3 -c "from raxtax_amd import _build; _build.build_lib()"
build() {  # name, flags
  local name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$ROOT/raxtax_amd/csrc" "$@" -x hip -c "$ROOT/raxtax_amd/csrc/rtx_hit_quad.hip" -o "$OUT/quad_$name.o"
  objs=$(ls "$ROOT"/raxtax_amd/_obj/*.o | grep -v rtx_hit_quad)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$OUT/lib_$name.so" $objs "$OUT/quad_$name.o" -lpthread
  echo "built $OUT/lib_$name.so"
}
buildp() {  # name, flags: a variant of rtx_hit_pair.hip
  local name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$ROOT/raxtax_amd/csrc" "$@" -x hip -c "$ROOT/raxtax_amd/csrc/rtx_hit_pair.hip" -o "$OUT/pair_$name.o"
  objs=$(ls "$ROOT"/raxtax_amd/_obj/*.o | grep -v rtx_hit_pair)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$OUT/lib_$name.so" $objs "$OUT/pair_$name.o" -lpthread
  echo "built $OUT/lib_$name.so"
}
buildk() {  # name, flags: a variant of rtx_kernels.hip
  local name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$ROOT/raxtax_amd/csrc" "$@" -x hip -c "$ROOT/raxtax_amd/csrc/rtx_kernels.hip" -o "$OUT/kern_$name.o"
  objs=$(ls "$ROOT"/raxtax_amd/_obj/*.o | grep -v "rtx_kernels.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$OUT/lib_$name.so" $objs "$OUT/kern_$name.o" -lpthread
  echo "built $OUT/lib_$name.so"
}
for v in "$@"; do
  case $v in
    base) build base ;;
    nomidfold) buildk nomidfold -DRTX_EXP_NO_MID_FOLD ;;
    sc1) buildk sc1 -DRTX_EXP_SC1_COUNT_STORES ;;
    seq4) buildp seq4 -DRTX_PAIRSEQ_NB=4 ;;
    seq6) buildp seq6 -DRTX_PAIRSEQ_NB=6 ;;
    ps1) buildp ps1 -DRTX_PAIR_STAMP=1 ;;
    ps2) buildp ps2 -DRTX_PAIR_STAMP=2 ;;
    ps3) buildp ps3 -DRTX_PAIR_STAMP=3 ;;
    ps4) buildp ps4 -DRTX_PAIR_STAMP=4 ;;
    pnt) buildp pnt -DRTX_EXP_NT_COUNT_STORES ;;
    pnohist) buildp pnohist -DRTX_EXP_NO_HIST ;;
    pnostore) buildp pnostore -DRTX_EXP_NO_COUNT_STORE ;;
    pnoboth) buildp pnoboth -DRTX_EXP_NO_COUNT_STORE -DRTX_EXP_NO_HIST ;;
    knosout) buildk knosout -DRTX_EXP_KMER_NO_SOUT ;;
    prune150) buildp prune150 -DRTX_EXP_PRUNE_EMU=150 ;;
    prune170) buildp prune170 -DRTX_EXP_PRUNE_EMU=170 ;;
    prune200) buildp prune200 -DRTX_EXP_PRUNE_EMU=200 ;;
    prune130) buildp prune130 -DRTX_EXP_PRUNE_EMU=130 ;;
    ks1) buildk ks1 -DRTX_KMER_STAMP=1 ;;
    ks2) buildk ks2 -DRTX_KMER_STAMP=2 ;;
    ks3) buildk ks3 -DRTX_KMER_STAMP=3 ;;
    ks4) buildk ks4 -DRTX_KMER_STAMP=4 ;;
    pskipb) buildp pskipb -DRTX_EXP_SKIP_EPI=1 ;;
    pskipab) buildp pskipab -DRTX_EXP_SKIP_EPI=2 ;;
    pskiploop) buildp pskiploop -DRTX_EXP_SKIP_LOOP ;;
    ps5) buildp ps5 -DRTX_PAIR_STAMP=5 ;;
    ps6) buildp ps6 -DRTX_PAIR_STAMP=6 ;;
    ps7) buildp ps7 -DRTX_PAIR_STAMP=7 ;;
    ps8) buildp ps8 -DRTX_PAIR_STAMP=8 ;;
    nb3) buildk nb3 -DRTX_HIT_NB=3 ;;
    nb4) buildk nb4 -DRTX_HIT_NB=4 ;;
    nb5) buildk nb5 -DRTX_HIT_NB=5 ;;
    nb6) buildk nb6 -DRTX_HIT_NB=6 ;;
    occ2) buildk occ2 -DRTX_EXP_HIT_LDS_KB=20 ;;
    occ3) buildk occ3 -DRTX_EXP_HIT_LDS_KB=13 ;;
    nw8) buildk nw8 -DRTX_PREFIX_NW=8 ;;
    nw2) buildk nw2 -DRTX_PREFIX_NW=2 ;;
    nw16) buildk nw16 -DRTX_PREFIX_NW=16 ;;
    p1) build p1 -DRTX_QUAD_AHEAD=1 ;;
    p2) build p2 -DRTX_QUAD_AHEAD=2 ;;
    nodma) build nodma -DRTX_QUAD_NO_DMA ;;
    w2) build w2 -DRTX_QUAD_WAVES_PER_SIMD=2 ;;
    st1) build st1 -DRTX_QUAD_STAMP=1 ;;
    st2) build st2 -DRTX_QUAD_STAMP=2 ;;
    st3) build st3 -DRTX_QUAD_STAMP=3 ;;
    st4) build st4 -DRTX_QUAD_STAMP=4 ;;
    st5) build st5 -DRTX_QUAD_STAMP=5 ;;
    st6) build st6 -DRTX_QUAD_STAMP=6 ;;
    st7) build st7 -DRTX_QUAD_STAMP=7 ;;
    rr16) build rr16 -DRTX_QUAD_RR=16 -DRTX_QUAD_AHEAD=1 ;;
    rr16w2) build rr16w2 -DRTX_QUAD_RR=16 -DRTX_QUAD_AHEAD=1 -DRTX_QUAD_WAVES_PER_SIMD=2 ;;
    rr32w2) build rr32w2 -DRTX_QUAD_RR=32 -DRTX_QUAD_AHEAD=1 -DRTX_QUAD_WAVES_PER_SIMD=1 ;;
    w2p4) build w2p4 -DRTX_QUAD_WAVES_PER_SIMD=2 -DRTX_QUAD_AHEAD=4 ;;
    w2p6) build w2p6 -DRTX_QUAD_WAVES_PER_SIMD=2 -DRTX_QUAD_AHEAD=6 ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
