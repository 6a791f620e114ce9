#!/bin/bash
# round 4, session z: the Schur tile kernel alone at the operand strides the factorisation gives it
B=scripts/micro/gemm_bench_d
for args in "7936 960 960 7936 9856" "7936 960 1000 8000 9920" "7936 960 8384 8384 8384" "7936 1024 1024 7936 9984" "7936 1024 1056 8000 10048" "12032 1536 1536 12032 15104" "12032 1536 1568 12096 15168" "12032 1536 12224 12224 12224"; do
  timeout 120 $B $args
done
