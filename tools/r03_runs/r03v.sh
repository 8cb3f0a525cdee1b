set -e
mkdir -p gpurun_out/r03v
bash tools/lib_sweep.sh gpurun_out/r03v/ab.txt "grid sphere instanced" $PWD/ab/lib_prev.so $PWD/ab/lib_cur.so $PWD/ab/lib_prev.so $PWD/ab/lib_cur.so
