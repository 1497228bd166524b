#!/bin/bash
# Ablation builds of the fused chain kernel (timing only: results are garbage by construction).  Run from the repo root in the
# authoring container to BUILD (hipcc cross-compiles), then on the GPU box with "run" to time them:
#   bash profiles/exp_chain_variants.sh build ; gpurun -- 'bash profiles/exp_chain_variants.sh run'
set -e
R=$PWD
V=$R/sttode_amd/lib/variants
if [ "$1" = build ]; then
  mkdir -p $V
  cd sttode_amd/csrc
  for name in ${VARIANTS:-base nodma nogather nogates nodma_nobarrier nodma_nogather_nogates all stamps st_nodma st_nogather st_nogates st_nodma_nogather st_all}; do
    case $name in
      stamps) D="-DC32_DIAG_STAMPS" ;; trace) D="-DC32_DIAG_TRACE" ;; trace_prio0) D="-DC32_DIAG_TRACE -DROLE_PRIO=0" ;; prio0) D="-DROLE_PRIO=0" ;; st_nodma) D="-DC32_DIAG_STAMPS -DC32_DIAG_NODMA" ;; st_nogather) D="-DC32_DIAG_STAMPS -DC32_DIAG_NOGATHER" ;;
      st_nogates) D="-DC32_DIAG_STAMPS -DC32_DIAG_NOGATES" ;; st_nodma_nogather) D="-DC32_DIAG_STAMPS -DC32_DIAG_NODMA -DC32_DIAG_NOGATHER" ;;
      st_all) D="-DC32_DIAG_STAMPS -DC32_DIAG_NODMA -DC32_DIAG_NOGATHER -DC32_DIAG_NOGATES -DC32_DIAG_NOBARRIER" ;;
      base) D="" ;; nodma) D="-DC32_DIAG_NODMA" ;; nogather) D="-DC32_DIAG_NOGATHER" ;; nogates) D="-DC32_DIAG_NOGATES" ;;
      nodma_nobarrier) D="-DC32_DIAG_NODMA -DC32_DIAG_NOBARRIER" ;; nodma_nogather_nogates) D="-DC32_DIAG_NODMA -DC32_DIAG_NOGATHER -DC32_DIAG_NOGATES" ;;
      all) D="-DC32_DIAG_NODMA -DC32_DIAG_NOGATHER -DC32_DIAG_NOGATES -DC32_DIAG_NOBARRIER" ;;
    esac
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $D -c chain32.hip -o /tmp/chain32_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib_$name.so /tmp/chain32_$name.o $(ls build/*.o | grep -v chain32.o)
  done
  exit 0
fi
mkdir -p gpurun_out/r02/variants
for S in 128 2048; do
  for name in base nodma nogather nogates nodma_nobarrier nodma_nogather_nogates all; do
    STTODE_HIP_LIB=$V/lib_$name.so python bench.py --legs none --no-cpu --serial --scenes $S --steps 8 --time-every 1 > gpurun_out/r02/variants/${name}_s$S.json 2>/dev/null || echo "$name $S failed"
  done
done
python - <<'PY'
import json, glob
for S in (128, 2048):
    for name in ('base', 'nodma', 'nogather', 'nogates', 'nodma_nobarrier', 'nodma_nogather_nogates', 'all'):
        try:
            d = json.load(open(f'gpurun_out/r02/variants/{name}_s{S}.json'))
            k = d['kernels']['trajectory_chain']
            print(f'scenes {S:5d} {name:24s} chain {k["mean_us"]:8.1f} us  {k["tflops"]:6.1f} TFLOP/s')
        except Exception as e:
            print(S, name, 'ERR', e)
PY
