"""slam_amd: MI355X-native front-end (matching, triangulation, PnP-RANSAC scoring)
for the KITTI stereo-SLAM pipeline, behind the reference's Python call signatures."""
__version__ = "0.1.0"
