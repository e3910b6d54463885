"""Data feed of the hot path: keeps a step's inputs on the GPU from the file to the loss.
device_feed: synthetic registration pairs generated on the device and batches drawn from a dataset resident in HBM.
disk_feed:   the reference's on-disk datasets (ModelNet40 h5 -> npz, FlyingThings3D npz): drop-in Dataset classes with the
             reference's `__getitem__`, and resident feeds that load the files once and serve whole batches from HBM."""
from .device_feed import RegistrationFeed, ResidentRegistrationFeed, uniform_clouds
from .disk_feed import (ClassificationData, ModelNet40Data, ResidentModelNet40, ResidentSceneflow, SceneflowDataset,
                        load_modelnet40, load_sceneflow_file)
