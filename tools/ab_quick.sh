#!/bin/bash
# usage: ab_quick.sh flight|ball variant...  ; env.step timing of the default library and each variant (tools/bench_flags.py), twice
kind=$1; shift
for i in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then L=""; else L=flybody_amd/csrc/variants/libflybody_env_$v.so; fi
  echo -n "$v: "; FLYBODY_ENV_LIB=$L timeout -k 10 150 python tools/bench_flags.py $kind 0 2>/dev/null | tail -1
done; done
