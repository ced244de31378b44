for v in slot rand; do for wl in dagnet_bsk_4096x20 avrnn_bsk_512x20 vrnn_bsk_4096x20; do
timeout -k 10 200 python tools/bench_lib.py dagnet_amd/_lib/exp/lib_$v.so --no-cpu-baseline --workload $wl > gpurun_out/E_${v}_$wl.json 2> gpurun_out/E_${v}_$wl.err || { tail -5 gpurun_out/E_${v}_$wl.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/E_${v}_$wl.json')); print('$v $wl kernel_ms %.2f frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
