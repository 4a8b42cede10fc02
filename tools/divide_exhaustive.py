"""Exhaustive proof run for nmf_opts.fast_divide = 1 inside the guard range: every pair of fp32 significands
(2^46 pairs, 64 launches) through quotient<1> and through the IEEE division, plus the exponent invariance of v_rcp_f32.
    gpurun -- python tools/divide_exhaustive.py [first_slice n_slices]      (stderr carries the counts)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmf_gpu_amd as ng
s = ng.Solver(256, 256, 64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s.time_piece(4002, 1)          # v_rcp_f32 exponent invariance
s.time_piece(4004, 1)          # harness self-check: x * rcp(y) must mismatch often
s.time_piece(4001, n)          # the shipped 6-instruction quotient: 0 mismatches
if len(sys.argv) > 2:
    s.time_piece(4003, n)      # the 4-instruction variant (no reciprocal refinement): 47 045 mismatches -- not usable
s.close()
