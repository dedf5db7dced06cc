import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ringsnark_amd import params as P
from ringsnark_amd.device import Device
d=Device(P.preset("C3"))
for _ in range(3): print(d.measure_peaks())
