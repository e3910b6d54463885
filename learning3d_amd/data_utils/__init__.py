"""Device-side data feed for the registration / forward benchmarks.  The reference's
data_utils (h5 / npz readers behind torch DataLoader workers) is out of scope: there are no datasets in this
environment; what is here keeps a step's inputs on the GPU from generation to loss."""
from .device_feed import RegistrationFeed, uniform_clouds
