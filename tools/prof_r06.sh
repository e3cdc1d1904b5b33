# Round-6 profiles of the build that is benchmarked.  Run on the GPU box from the repository root:
#   bash tools/prof_r06.sh [steps...]   -> gpurun_out/prof_r06/*   (copy the summaries into profiles/r06/)
# Every rocprofv3 call has the program itself after `--`; --pmc passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/prof_r06; mkdir -p $O
STEPS="${*:-0 1 2 3 4 5}"
has() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
BENCH="--steps 1 --warmup 1 --no-cpu-baseline --no-other-modes --no-configs"
if has 0; then
echo "== 0 the bench line itself, default arguments (same box as everything below)"
timeout -k 10 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
fi
if has 1; then
echo "== 1 kernel-trace stats, gates mode (bench.py $BENCH)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py $BENCH > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
cp $(find $O/bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
fi
if has 2; then
echo "== 2 kernel-trace stats of k_pbs64k in the 46-bit pair (2,048 two-input LUTs, the binary's LUT-mode set) and in the 49-bit pair"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lut_m1c1 -o lut_m1c1 -- python3 tools/prof_luts.py 2048 3 shortint_m1c1 2 > $O/lut_m1c1.log 2>&1 || tail -5 $O/lut_m1c1.log
cp $(find $O/lut_m1c1 -name "*kernel_stats.csv" | head -1) $O/lut_m1c1_kernel_stats.csv
export HELM_SI_FIELD=49
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lut_m1c1_49 -o lut_m1c1 -- python3 tools/prof_luts.py 2048 3 shortint_m1c1 2 > $O/lut_m1c1_49.log 2>&1 || tail -5 $O/lut_m1c1_49.log
unset HELM_SI_FIELD
cp $(find $O/lut_m1c1_49 -name "*kernel_stats.csv" | head -1) $O/lut_m1c1_field49_kernel_stats.csv
fi
if has 3; then
echo "== 3 world 8 as eight rank threads on this one GPU (HELM_BENCH_REHEARSE=threads): the N > 1 record with its plan, the N = 1 job inside the run, every row's baseline"
HELM_BENCH_REHEARSE=threads timeout -k 10 900 python3 bench.py --gpus 8 --blocks 8 --steps 1 --warmup 1 > $O/rehearse_n8.json 2> $O/rehearse_n8.err || tail -5 $O/rehearse_n8.err
grep "^\[bench\]" $O/rehearse_n8.err
fi
if has 4; then
echo "== 4 the exchange machinery on one GPU: bench.py --force-comm (world-size-1 RCCL communicator inside the library)"
timeout -k 10 400 python3 bench.py --force-comm --no-cpu-baseline --no-other-modes --no-configs > $O/bench_n1_forcecomm.json 2> $O/bench_fc.err || tail -5 $O/bench_fc.err
fi
if has 5; then
echo "== 5 the driver's arguments"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_steps20.json 2> $O/bench_n1_steps20.err || tail -5 $O/bench_n1_steps20.err
fi
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_r06/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d.get("roofline", {})
    print(f.split("/")[-1], d.get("value"), d.get("headline_kind"), "frac", r.get("frac"), "single", (d.get("single_block") or {}).get("wall_s"),
          "m1c1", ((d.get("other_modes") or {}).get("lut_mode_m1c1") or {}).get("roofline", {}).get("frac"), "power", (r.get("power_and_clock_over_the_timed_steps") or {}).get("socket_power_w"),
          "err", d.get("error"))
PY
head -3 $O/bench_kernel_stats.csv $O/lut_m1c1_kernel_stats.csv $O/lut_m1c1_field49_kernel_stats.csv 2>/dev/null | cut -c1-260
