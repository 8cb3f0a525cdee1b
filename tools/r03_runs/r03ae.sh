set -e
bash tools/pmc_passes.sh r3_instanced_1x2000000 --config instanced --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_cornell_1x2000000 --config cornell --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_3x2000000 --config instanced --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_cornell_3x750000 --config cornell --steps 1 --warmup 1
