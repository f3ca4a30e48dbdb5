"""Development aid: verbose interior-point trace of one mesh iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycollo_amd import problems
from pycollo_amd.iteration import MeshIteration
name = sys.argv[1]; kw = eval(sys.argv[2]) if len(sys.argv) > 2 else {}
it = MeshIteration(problems.REGISTRY[name](**kw))
res = it.solve_with_ipm(verbose=1, max_iter=int(os.environ.get("MAXIT", "400")))
print(res.status, res.iterations, it.objective, res.evaluations)
