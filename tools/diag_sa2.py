import torch, sys
sys.argv = ["x"]
import tests.test_gpu_float as T
for i in range(3):
    try:
        T.test_sa_ball_query_fps_and_global_sa_vs_oracle()
        print("run", i, "ok")
    except AssertionError as e:
        print("run", i, "FAIL", str(e)[:120])
