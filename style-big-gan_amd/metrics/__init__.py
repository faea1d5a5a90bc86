"""Quality-metric plumbing: feature statistics over the data set and over generated images, and the metrics
computed from them.  Interface of the reference's ``stylegan2ada/metrics`` package (``metric_main.calc_metric / report_metric``,
``metric_utils.MetricOptions / FeatureStats / compute_feature_stats_for_*``).  The reference fetches its feature detectors (Inception,
VGG16 TorchScript files) from a URL; nothing is downloaded here -- a detector is a LOCAL TorchScript file or any callable
``images uint8 [N, 3, H, W] -> features [N, F]`` handed in through ``MetricOptions(detector=...)``."""
