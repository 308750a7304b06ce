#!/bin/bash
# ONE parameterised runner for a GPU-box session (replaces the per-session gpu_r*.sh scripts of rounds 2-3):
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh TAG STEP [STEP ...]'
# Everything lands in gpurun_out/TAG/.  Steps (each under its own timeout so that a hang cannot eat the box):
#   tests[:K]            pytest -m gpu (optionally -k K)                      -> tests.log
#   smoke                __graft_entry__.smoke()                              -> smoke.log
#   bench[:ARGS]         python bench.py ARGS                                 -> bench.json / bench.err
#   kt                   rocprofv3 --kernel-trace of a short bench            -> kernel_stats.txt
#   pmc                  FETCH_SIZE / WRITE_SIZE passes + gather calibration  -> pmc_traffic.json, pmc_*.txt
#                        (needs csrc/ubench/ubench.bin: hipcc --offload-arch=gfx950 -O3 -o ubench.bin ubench.hip)
#   pmcpy:SCRIPT[:ARGS]  FETCH_SIZE / WRITE_SIZE passes of python tools/SCRIPT ARGS   -> pmcpy_SCRIPT.txt
#   sq                   SQ counters (VALU / wait / busy) of a short bench    -> pmc_sq.txt
#   env:VAR=VAL          export VAR=VAL for the steps that follow (A/B switches: ARK_HIP_FFT_LAZY=0, ARK_HIP_MSM_LAZY=0)
#   n2gloo               bench.py --gpus 2 over gloo, ranks sharing the GPU   -> bench_n2_gloo.json
#   soak:N               tools/soak.py N                                      -> soak.log
#   skewsoak:N[:SEED[:LO:HI]]  tools/skew_soak.py N SEED [LO HI] (random width-class mixtures, five curves; sizes
#                        2^LO..2^HI instead of 2^17..2^21)                    -> skew_soak.log (appended)
#   ktpy:SCRIPT[:ARGS]   rocprofv3 --kernel-trace of python tools/SCRIPT ARGS; KT_TIMELINE=N also lists the last N launches
#                        in start order with the idle gaps between them        -> kernel_stats_SCRIPT_ARGS.txt
#   pyv:SCRIPT[:ARGS]    python tools/SCRIPT on the shipped library and on every algebra_amd/variants/*.so -> pyv_SCRIPT.txt
#   mulbench             csrc/ubench/mulbench_*.bin (prebuilt, travel as .bin)-> mulbench.txt
#   msm:CURVE:LOGN[:MODE[:REPS]]   tools/msm_bench.py, ARK_HIP_MSM_LAZY=1 and 0 -> msm.txt
#   fft[:LO:HI]          tools/fft_shapes.py (the reference's five bench shapes), ARK_HIP_FFT_LAZY=1 and 0 -> fft.txt
#   fftv[:LO:HI]         tools/fft_shapes.py on the shipped library and every algebra_amd/variants/*.so -> fftv.txt
#   ab:CURVE:LOGN[:MODE] every algebra_amd/variants/*.so x LAZY in {1,0}, twice -> ab.txt
#   py:SCRIPT[:ARGS]     python tools/SCRIPT ARGS (':' separates arguments)   -> py_SCRIPT.txt
tag=$1; shift
mkdir -p gpurun_out/$tag
O=$PWD/gpurun_out/$tag
R=$PWD
export TMPDIR=/tmp
db() { find $O/$1 -name "*results.db" | head -1; }
SHORT="python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras --fft-steps 3"
for step in "$@"; do
  IFS=':' read -r -a a <<< "$step"
  case ${a[0]} in
    tests)
      if [ -n "${a[1]}" ]; then (timeout 1200 python -m pytest tests -m gpu -q -x -k "${a[1]}" 2>&1 | tail -25) > $O/tests_${a[1]// /_}${suffix}.log
      else (timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/tests.log; fi ;;
    smoke) (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log) 2>&1 ;;
    bench) (timeout 900 python bench.py ${a[@]:1} > $O/bench.json) 2> $O/bench.err ;;
    kt)
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace -d $O/p_kt -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.out 2> $O/kt.err
      cd $R
      python tools/rocpd_stats.py $(db p_kt) --min-us 100 > $O/kernel_stats.txt 2>> $O/post.err
      rm -rf $O/p_kt ;;
    ktpy)   # ktpy:SCRIPT[:ARGS]: kernel trace of python tools/SCRIPT ARGS -> kernel_stats_SCRIPT_ARGS.txt
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace -d $O/p_ktpy -o kt -- python $R/tools/${a[1]} ${a[@]:2} >> $O/ktpy.out 2>> $O/ktpy.err
      cd $R
      python tools/rocpd_stats.py $(db p_ktpy) --min-us 20 --timeline ${KT_TIMELINE:-0} > $O/kernel_stats_${a[1]%.py}_$(echo ${a[@]:2} | tr ' ' '_')${suffix}.txt 2>> $O/post.err
      rm -rf $O/p_ktpy ;;
    pmc)
      cd /tmp
      timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/p_fetch -o f -- $SHORT > $O/fetch.out 2> $O/fetch.err
      timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/p_write -o w -- $SHORT > $O/write.out 2> $O/write.err
      timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/p_cal -o c -- $R/algebra_amd/csrc/ubench/ubench.bin > $O/cal.out 2> $O/cal.err
      cd $R
      python tools/rocpd_stats.py $(db p_fetch) --pmc --min-us 100 > $O/pmc_fetch.txt 2>> $O/post.err
      python tools/rocpd_stats.py $(db p_write) --pmc --min-us 100 > $O/pmc_write.txt 2>> $O/post.err
      python tools/pmc_traffic.py $(db p_fetch) $(db p_write) 24 22 $(db p_cal) 33554432 > $O/pmc_traffic.json 2>> $O/post.err
      rm -rf $O/p_fetch $O/p_write $O/p_cal ;;
    pmcpy)   # pmcpy:SCRIPT[:ARGS]: FETCH_SIZE and WRITE_SIZE passes (separate runs) of python tools/SCRIPT ARGS -> pmcpy_SCRIPT<suffix>.txt
      cd /tmp
      timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/p_pf -o f -- python $R/tools/${a[1]} ${a[@]:2} > $O/pmcpy.out 2> $O/pmcpy.err
      timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/p_pw -o w -- python $R/tools/${a[1]} ${a[@]:2} >> $O/pmcpy.out 2>> $O/pmcpy.err
      cd $R
      (echo "== ${a[@]:1} $suffix (raw counters, KiB; FETCH_SIZE counts half of a wide streaming read on gfx950)"
       python tools/rocpd_stats.py $(db p_pf) --pmc --min-us 50; python tools/rocpd_stats.py $(db p_pw) --pmc --min-us 50) >> $O/pmcpy_${a[1]%.py}$suffix.txt 2>> $O/post.err
      rm -rf $O/p_pf $O/p_pw ;;
    sq)
      cd /tmp
      timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/p_sq -o s -- $SHORT > $O/sq.out 2> $O/sq.err
      cd $R
      python tools/rocpd_stats.py $(db p_sq) --pmc --min-us 50 > $O/pmc_sq$suffix.txt 2>> $O/post.err
      rm -rf $O/p_sq ;;
    n2gloo)
      (ARK_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 --log-total 22 --fft-log-n 20 --fft-steps 4 > $O/bench_n2_gloo.json) 2> $O/bench_n2_gloo.err ;;
    soak) (timeout 1200 python tools/soak.py ${a[1]:-200} 2>&1 | tail -4) > $O/soak.log ;;
    skewsoak) (timeout 1500 python tools/skew_soak.py ${a[1]:-100} ${a[2]:-2024} ${a[3]} ${a[4]} 2>&1 | grep -v "^it " | tail -8) >> $O/skew_soak.log ;;
    mulbench)
      for b in algebra_amd/csrc/ubench/mulbench_*.bin; do (echo "== $b"; timeout 120 $b) >> $O/mulbench.txt 2>> $O/mulbench.err; done ;;
    msm)
      for lz in 1 0; do
        (echo "== ARK_HIP_MSM_LAZY=$lz ARK_HIP_MSM_GROUPS=$ARK_HIP_MSM_GROUPS"; ARK_HIP_MSM_LAZY=$lz timeout 600 python tools/msm_bench.py ${a[1]} ${a[2]} ${a[4]:-3} ${a[3]:-plain}) >> $O/msm.txt 2>> $O/msm.err
      done ;;
    fft)
      for lz in 1 0; do
        (echo "== ARK_HIP_FFT_LAZY=$lz"; ARK_HIP_FFT_LAZY=$lz timeout 600 python tools/fft_shapes.py ${a[1]:-16} ${a[2]:-24}) >> $O/fft.txt 2>> $O/fft.err
      done ;;
    pyv)   # pyv:SCRIPT[:ARGS]: python tools/SCRIPT on the shipped library and on every algebra_amd/variants/*.so
      for v in algebra_amd/libark_hip.so $(ls algebra_amd/variants/*.so); do
        (echo "== $v ${a[@]:1}"; ARK_HIP_LIB=$PWD/$v timeout 900 python tools/${a[1]} ${a[@]:2}) >> $O/pyv_${a[1]%.py}.txt 2>> $O/pyv_${a[1]%.py}.err
      done ;;
    fftv)
      for v in algebra_amd/libark_hip.so $(ls algebra_amd/variants/*.so); do
        (echo "== $v"; ARK_HIP_LIB=$PWD/$v timeout 600 python tools/fft_shapes.py ${a[1]:-20} ${a[2]:-24}) >> $O/fftv.txt 2>> $O/fftv.err
      done ;;
    ab)
      for rep in 1 2; do for v in $(ls algebra_amd/variants/*.so); do for lz in 1 0; do
        (echo "== $v LAZY=$lz ${a[1]} ${a[2]}"; ARK_HIP_MSM_LAZY=$lz ARK_HIP_LIB=$PWD/$v timeout 300 python tools/msm_bench.py ${a[1]} ${a[2]} 3 ${a[3]:-plain}) >> $O/ab.txt 2>> $O/ab.err
      done; done; done ;;
    py) (echo "== ${a[@]:1}"; timeout 900 python tools/${a[1]} ${a[@]:2}) >> $O/py_${a[1]%.py}.txt 2>> $O/py_${a[1]%.py}.err ;;
    env) export "${a[1]}"; suffix="_${a[1]//=/_}" ;;   # env:VAR=VAL -- exported for the steps that follow
    *) echo "unknown step $step" >> $O/errors.log ;;
  esac
  echo "$step done $(date +%s)" >> $O/steps.log
done
echo done > $O/done
