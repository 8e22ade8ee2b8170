#!/bin/bash
# cache-policy bits on the absorb's half-stencil atomics (timing builds build/libwiski_scatmodN.so: 1 sc1, 2 nt, 3 sc1 nt): does any of them leave
# A_h better placed for the SpMVs that follow?  bench trace: SpMV by position in the solve, the absorb kernel, updates/s
# build the variants first (on the build host):  for m in 1 2 3; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWISKI_SCATTER_ATOMIC_MOD=$m -c online_gp_amd/csrc/scatter_stats.hip -o build/scat_mod$m.o &&
#   hipcc --offload-arch=gfx950 -fPIC -shared -o build/libwiski_scatmod$m.so build/scat_mod$m.o $(ls build/obj/*.o | grep -v scatter_stats) -ldl; done
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4scatmod; mkdir -p $O; rm -f $O/out.txt
cd /tmp
for m in 0 1 2 3 0; do
  rm -rf /tmp/prof_b
  if [ $m = 0 ]; then unset WISKI_HIP_SO; else export WISKI_HIP_SO=$R/build/libwiski_scatmod$m.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --blocks 8 > $O/prof.log 2>&1
  echo "== atomic mod $m" >> $O/out.txt
  python $R/tools/spmv_trace_split.py /tmp/prof_b/bench_kernel_trace.csv | grep "^   [012] \|^all" >> $O/out.txt
  grep "k_scatter_stats_sym" /tmp/prof_b/bench_kernel_stats.csv | awk -F'","|",|,' '{print "scatter avg ns", $(NF-4)}' >> $O/out.txt
  grep -o '"value": [0-9.]*' $O/prof.log | head -1 >> $O/out.txt
done
cat $O/out.txt
