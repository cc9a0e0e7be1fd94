mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
for mode in fused streamed; do
  O=$R/gpurun_out/traffic_$mode; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python $R/tools/fwd_only.py 68 $mode 4 > $O/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python $R/tools/fwd_only.py 68 $mode 4 > $O/write.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib_fetch -o p --output-format csv -- python $R/tools/pmc_calib.py > $O/cf.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib_write -o p --output-format csv -- python $R/tools/pmc_calib.py > $O/cw.log 2>&1
  python $R/tools/pmc_traffic.py $O 5 $R/gpurun_out/traffic_$mode.json
done
