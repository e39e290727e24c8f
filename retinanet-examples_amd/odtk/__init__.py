"""odtk -- MI355X-native drop-in for the post-processing path of NVIDIA/retinanet-examples."""
