# the round's closing session: full GPU suite, then tools/measure_round.sh (both logs under gpurun_out/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $R/gpurun_out/r04_full_gputest_d.log 2>&1
rc=$?
tail -3 $R/gpurun_out/r04_full_gputest_d.log
[ $rc -eq 0 ] || exit $rc
bash tools/measure_round.sh > $R/gpurun_out/measure_round_r04d.log 2>&1
tail -5 $R/gpurun_out/measure_round_r04d.log
