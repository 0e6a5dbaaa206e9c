"""Import-name shim: only `mmdet.datasets.pipelines.Compose` exists."""
