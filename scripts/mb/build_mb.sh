#!/bin/bash
# Cross-compiles the MLP-loop micro-benchmark variants quoted in DESIGN.md section 4 (hipcc, no GPU needed);
# run them on an MI355X with scripts/mb/run_mb.sh.  Binaries land in scripts/mb/bin/ (git-ignored).
set -u
cd "$(dirname "$0")"
mkdir -p bin
build() { name=$1; src=$2; shift 2; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVARIANT="\"$name\"" "$@" "$src" -o "bin/$name" || exit 1; }
# structure of the shipped kernel and what each element of it costs
build a0_shipped        mb_mlp.hip -DPRIO -DLDSPAD=90000 &
build a1_no_dma         mb_mlp.hip -DPRIO -DNODMA -DLDSPAD=90000 &
build a2_no_dma_no_bar  mb_mlp.hip -DNODMA -DNOBAR -DLDSPAD=90000 &
build a3_epilogue60     mb_mlp.hip -DPRIO -DEPI=60 -DLDSPAD=90000 &
wait
# alternatives that were measured and not adopted
build b0_split_k_acc    mb_mlp.hip -DPRIO -DDUAL -DLDSPAD=90000 &
build b1_dma_by_prio    mb_mlp.hip -DPRIO -DDMAHI -DLDSPAD=90000 &
build b2_staggered_dma  mb_mlp.hip -DPRIO -DSTAG -DLDSPAD=90000 &
build b3_16kb_chunks    mb_mlp.hip -DPRIO -DPIECES16 -DLDSPAD=90000 &
wait
build b4_ring_depth8    mb_mlp.hip -DPRIO -DAFD=8 -DLDSPAD=90000 &
build b5_vgpr_staged    mb_mlp.hip -DPRIO -DVLOAD -DLDSPAD=90000 &
build b6_rotated_pieces mb_mlp.hip -DPRIO -DROT -DLDSPAD=90000 &
build b7_half_phase     mb_half.hip -DPRIO -DDMAHI &
build c0_endbar         mb_mlp.hip -DENDBAR -DLDSPAD=90000 &
build c1_endbar_prio    mb_mlp.hip -DENDBAR -DPRIO -DLDSPAD=90000 &
build c2_endbar_epi60   mb_mlp.hip -DENDBAR -DEPI=60 -DLDSPAD=90000 &
build d0_bar2          mb_mlp.hip -DPRIO -DBAR2 -DLDSPAD=80000 &
build d1_bar2_epi60    mb_mlp.hip -DPRIO -DBAR2 -DEPI=60 -DLDSPAD=80000 &
build d2_bar2_nodma    mb_mlp.hip -DPRIO -DBAR2 -DNODMA -DLDSPAD=80000 &
wait
ls bin
# round 2: two independent 4-wave workgroups per CU (bin4/)
mkdir -p bin4
b4() { n=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVARIANT="\"$n\"" "$@" -o bin4/$n || exit 1; }
b4 g0_a0_8w mb_mlp.hip -DPRIO -DRANDIMG -DLDSPAD=90000 &
b4 g0_b7_half mb_half.hip -DPRIO -DDMAHI -DRANDIMG &
b4 g1_ring3 mb_wg4.hip -DRING3 -DRANDIMG -DLDSPAD=28000 &
b4 g2_ring2 mb_wg4.hip -DRANDIMG -DLDSPAD=45000 &
wait
b4 g4_ring2_epi60 mb_wg4.hip -DRANDIMG -DEPI=60 -DLDSPAD=45000 &
b4 g5_a3_epi60_8w mb_mlp.hip -DPRIO -DRANDIMG -DEPI=60 -DLDSPAD=90000 &
b4 g6_ring2_nodma_nobar mb_wg4.hip -DRANDIMG -DNODMA -DNOBAR -DLDSPAD=45000 &
b4 g8_ring3_epi60 mb_wg4.hip -DRING3 -DRANDIMG -DEPI=60 -DLDSPAD=28000 &
wait
