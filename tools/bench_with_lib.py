import sys, os
sys.path.insert(0, "/root/repo")
from dae_rnn_news_recommendation_amd import _lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
