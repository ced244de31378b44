for kw in '{"n_layers": 1, "graph_model": "gcn"}' '{"n_layers": 2, "graph_model": "gcn"}' '{"n_layers": 1}' '{"n_layers": 2}' '{"n_layers": 2, "z_dim": 16, "graph_model": "gcn"}' '{"n_layers": 2, "z_dim": 16}'; do
  KW="$kw" timeout -k 10 100 python tools/diag_kw11.py 2>&1 | grep -v amdgpu
done
echo ---- splitc; for kw in '{"n_layers": 1, "graph_model": "gcn"}' '{"n_layers": 2, "graph_model": "gcn"}'; do KW="$kw" DAGNET_HIP_LIB=$PWD/dagnet_amd/_lib/libdagnet_hip_splitc.so timeout -k 10 100 python tools/diag_kw11.py 2>&1 | grep -v amdgpu; done
echo ---- m64; KW='{"n_layers": 1, "graph_model": "gcn"}' DAGNET_HIP_LIB=$PWD/dagnet_amd/_lib/libdagnet_hip_m64.so timeout -k 10 100 python tools/diag_kw11.py 2>&1 | grep -v amdgpu
