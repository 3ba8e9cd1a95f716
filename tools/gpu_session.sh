#!/bin/bash
# One parametrised GPU session (replaces the per-session tools/gpu_round*_*.sh scripts of rounds 3 / 4):
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'     -> gpurun_out/<tag>/...
# steps (run in the order given):
#   tests            the whole GPU suite (pytest -m gpu)
#   ab:<lib,...>     tools/ab.py on each library (product = "default"; others by the tag of serl_amd/csrc/libserl_amd_<tag>.so), twice each,
#                    150 episodes of the SERL50 shape; AB_E / AB_ACTORS override
#   bench            the driver's command (bench.py, defaults)
#   bench:<args>     bench.py with the given arguments (commas for spaces), e.g. bench:--workload,serl10,--no-cpu-baseline
#   pmc              the SQ issue counters + FETCH / WRITE sizes of one evaluation (separate passes, MI355X_MICROARCH.md)
#   icache:<args>    instruction-cache counters (SQC_ICACHE_REQ / HITS / MISSES / MISSES_DUPLICATE, SQ_IFETCH) of one evaluation of bench.py <args>
#   pcsamp:<lib>:<args>   PC sampling (rocprofv3 --pc-sampling-beta-enabled) of bench.py <args> --steps 1 --warmup 0 on library <lib> (default | the tag of
#                    libserl_amd_<tag>.so; `g` = the product's code with line tables): tools/pcsamp.py collect -> <tag>/pcsamp_<lib>_<args>.json
#   ldspmc:<lib,...> LDS counters (SQ_LDS_BANK_CONFLICT / ADDR_CONFLICT / IDX_ACTIVE / UNALIGNED_STALL, SQ_INSTS_LDS, SQ_ACTIVE_INST_LDS, SQ_WAIT_INST_LDS) of one
#                    evaluation of the default bench on each library
#   roleprof:<lib,...>   instruction / wait counters of one launch per library (ablation builds: tools/isa/role_profile.py)
#   saturate[:args]  tools/bench_saturate.py (SURVEY 8d's saturating configuration)
#   profile          tools/profile_round.sh <tag> (the round's whole profile series)
#   py:<script.py>[,arg,...]   any script of the repo
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for STEP in "$@"; do
  case $STEP in
    tests)
      timeout 1500 python -m pytest tests -x -q -m gpu --timeout=600 > $O/pytest_gpu.txt 2>&1
      tail -n 4 $O/pytest_gpu.txt ;;
    ab:*)
      for L in $(echo ${STEP#ab:} | tr ',' ' '); do
        for rep in 1 2; do
          if [ $L = default ]; then timeout 300 python tools/ab.py ${AB_E:-150} >> $O/ab.txt 2>> $O/err.txt
          else SERL_LIB=$R/serl_amd/csrc/libserl_amd_$L.so timeout 300 python tools/ab.py ${AB_E:-150} >> $O/ab.txt 2>> $O/err.txt; fi
        done
      done
      cut -c1-200 $O/ab.txt ;;
    bench)
      timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json ;;
    bench:*)
      A=$(echo ${STEP#bench:} | tr ',' ' '); N=$(echo ${STEP#bench:} | tr -c 'a-zA-Z0-9' '_')
      timeout 900 python bench.py $A > $O/bench$N.json 2> $O/bench$N.err; tail -c 600 $O/bench$N.json ;;
    pmc)
      python serl_amd/build.py --source-hash > $O/csrc_sha256.txt
      (cd /tmp && export TMPDIR=/tmp
       P1="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
       timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pf -- $P1 > $O/pmc_fetch.log 2>&1
       timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pw -- $P1 > $O/pmc_write.log 2>&1
       timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmc_sq -o ps -- $P1 > $O/pmc_sq.log 2>&1)
      for k in fetch write sq; do python tools/pmc_summary.py $O/pmc_$k > $O/pmc_$k.json 2>> $O/err.txt; rm -rf $O/pmc_$k; done
      cat $O/pmc_sq.json | cut -c1-600 ;;
    icache:*)
      A=$(echo ${STEP#icache:} | tr ',' ' '); N=$(echo ${STEP#icache:} | tr -c 'a-zA-Z0-9' '_')
      (cd /tmp && export TMPDIR=/tmp
       timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/pmc_ic$N -o pi -- python $R/bench.py $A --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_ic$N.log 2>&1)
      python tools/pmc_summary.py $O/pmc_ic$N > $O/pmc_ic$N.json 2>> $O/err.txt; rm -rf $O/pmc_ic$N
      tr -d '\n' < $O/pmc_ic$N.json | cut -c1-500; echo ;;
    pcsamp:*)
      REST=${STEP#pcsamp:}; L=${REST%%:*}; A=$(echo ${REST#*:} | tr ',' ' '); N=$(echo ${REST} | tr -c 'a-zA-Z0-9' '_')
      LIBARG=""; [ $L != default ] && LIBARG="--lib $R/serl_amd/csrc/libserl_amd_$L.so"
      timeout 2400 python tools/pcsamp.py collect $O/pcsamp_$N.json $LIBARG -- python $R/bench.py $A --steps 1 --warmup 0 --no-cpu-baseline > $O/pcsamp_$N.log 2>&1
      tail -n 8 $O/pcsamp_$N.log | cut -c1-400 ;;
    ldspmc:*)
      for L in $(echo ${STEP#ldspmc:} | tr ',' ' '); do
        LIBENV=""; [ $L != default ] && LIBENV="SERL_LIB=$R/serl_amd/csrc/libserl_amd_$L.so"
        (cd /tmp && export TMPDIR=/tmp
         P1="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
         env $LIBENV timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $O/pmc_lds_$L -o pl -- $P1 > $O/pmc_lds_$L.log 2>&1
         env $LIBENV timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_LDS -d $O/pmc_sq_$L -o ps -- $P1 > $O/pmc_sq_$L.log 2>&1)
        for k in lds sq; do python tools/pmc_summary.py $O/pmc_${k}_$L > $O/pmc_${k}_$L.json 2>> $O/err.txt; rm -rf $O/pmc_${k}_$L; done
        echo $L; tr -d '\n' < $O/pmc_lds_$L.json | cut -c1-600; echo
      done ;;
    roleprof:*)
      # hardware instruction counters of ONE launch (150 episodes x 2 001 steps) per library: the ablation builds of tools/sweeps/r06_ablate.json against `frz`
      # (tools/isa/role_profile.py --pmc <tag>/roleprof.json turns the differences into instructions per role and env step)
      echo "{" > $O/roleprof.json
      for L in $(echo ${STEP#roleprof:} | tr ',' ' '); do
        LIBENV=""; [ $L != default ] && LIBENV="SERL_LIB=$R/serl_amd/csrc/libserl_amd_$L.so"
        (cd /tmp && export TMPDIR=/tmp
         P1="python $R/tools/one_rollout.py"
         env $LIBENV timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 -d $O/rp1_$L -o p -- $P1 > $O/rp1_$L.log 2>&1
         env $LIBENV timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SENDMSG -d $O/rp2_$L -o p -- $P1 > $O/rp2_$L.log 2>&1
         env $LIBENV timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS -d $O/rp3_$L -o p -- $P1 > $O/rp3_$L.log 2>&1)
        echo "\"$L\": [" >> $O/roleprof.json
        for k in 1 2 3; do python tools/pmc_summary.py $O/rp${k}_$L >> $O/roleprof.json 2>> $O/err.txt; [ $k != 3 ] && echo "," >> $O/roleprof.json; rm -rf $O/rp${k}_$L; done
        echo "]," >> $O/roleprof.json
      done
      echo "\"episodes\": 150, \"steps_per_episode\": 2001}" >> $O/roleprof.json
      python -c "import json; d=json.load(open('$O/roleprof.json')); print({k: round(v[0].get('SQ_INSTS_VALU', 0) / 300150) for k, v in d.items() if isinstance(v, list)})" ;;
    saturate*)
      A=$(echo ${STEP#saturate} | tr ':,' '  ')
      timeout 1500 python tools/bench_saturate.py $A > $O/saturate.jsonl 2>> $O/err.txt; cut -c1-420 $O/saturate.jsonl ;;
    profile)
      bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1 ;;
    py:*)
      A=$(echo ${STEP#py:} | tr ',' ' '); S1=${STEP#py:}; S1=${S1%%,*}
      timeout 900 python $A > $O/$(basename $S1 .py).txt 2>> $O/err.txt; tail -n 5 $O/$(basename $S1 .py).txt | cut -c1-1200 ;;
  esac
done
[ -f $O/err.txt ] && tail -n 3 $O/err.txt
true
