# All counter passes this round's roofline blocks read: headline shape and exclusive one-pool shape, reference-exact and tolerance mode.
set -e
bash tools/pmc_passes.sh r3_sphere_3x750000 --config sphere --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_sphere_1x2000000 --config sphere --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_grid_1x2000000 --config grid --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_grid_4x1500000 --config grid --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_grid_1x2000000_fast --config grid --fast --check-period 0 --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_grid_4x1500000_fast --config grid --fast --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_1x2000000 --config instanced --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_4x1500000 --config instanced --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_ext_1x2000000 --config instanced_ext --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_ext_4x1500000 --config instanced_ext --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_cornell_1x2000000 --config cornell --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_cornell_3x750000 --config cornell --steps 1 --warmup 1
# then, per shape:  python3 tools/pmc_roofline.py r03 <config> <shape>[_fast] gpurun_out/pmc_r3_<config>_<shape>[_fast] profiles/r03_valu_mix_<kernel>.txt
