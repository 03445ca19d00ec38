cat > /tmp/ab.py <<'PY'
import torch, sys, os
sys.path.insert(0,'.')
import tf_geometric_amd as tfg
from tf_geometric_amd.plan import gemm_bias_act
import bench
for (M,K,N) in [(173312,1433,16),(233000,602,16),(233000,602,64),(2708,1433,16),(170000,1433,256),(233000,602,8),(100000,301,40),(2400000,100,300),(50000,7,5),(2400000,256,300)]:
    a=torch.randn(M,K,device='cuda'); b=torch.randn(K,N,device='cuda')
    ts=sorted(bench._time(lambda: gemm_bias_act(a,b), steps=20, warmup=5) for _ in range(5))
    tt=sorted(bench._time(lambda: a@b, steps=20, warmup=5) for _ in range(3))
    ref=(a[:20000].double()@b.double())
    e1=float((gemm_bias_act(a,b)[:20000].double()-ref).abs().max())
    print(os.environ.get("TFGX_GEMM_UNALIGNED_V4","1"),M,K,N,"tfgx %.3f ms (min %.3f) torch %.3f maxerr %.2e"%(ts[2],ts[0],tt[1],e1))
PY
for i in 1; do TFGX_GEMM_UNALIGNED_V4=0 python /tmp/ab.py; TFGX_GEMM_UNALIGNED_V4=1 python /tmp/ab.py; done
python -m pytest tests/test_gpu_layers.py -q -x -k "gemm" 2>&1 | tail -2
