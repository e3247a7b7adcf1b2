defmodule NxSignalAMD.NIF do
  @moduledoc false
  # Thin loader of nif/nxsig_nif.c.  Every function is replaced at load time; the bodies below only run when the
  # shared object is missing.  Names and arities must equal the `funcs[]` table of the shim
  # (tests/test_nif_shim.py::test_nif_ex_matches_the_shim_table checks both directions).
  @on_load :load_nif

  def load_nif do
    # NXSIG_NIF_PATH (the shared object, with or without ".so") wins over the application's priv directory: what `elixir smoke.exs`
    # uses, where there is no :nx_signal_amd application (and no priv_dir) at all
    path =
      case System.get_env("NXSIG_NIF_PATH") do
        p when is_binary(p) and p != "" -> p |> Path.rootname(".so") |> String.to_charlist()
        _ -> :filename.join(:code.priv_dir(:nx_signal_amd), ~c"nxsig_nif")
      end

    :erlang.load_nif(path, 0)
  end

  def device_count(), do: :erlang.nif_error(:nif_not_loaded)
  def ctx_create(_device), do: :erlang.nif_error(:nif_not_loaded)
  def sync(_ctx), do: :erlang.nif_error(:nif_not_loaded)
  def last_dispatch(_ctx), do: :erlang.nif_error(:nif_not_loaded)
  def window(_kind, _n, _periodic, _beta, _eps), do: :erlang.nif_error(:nif_not_loaded)
  def firwin(_taps, _cutoff, _kind, _beta, _pass_zero, _scale, _fs), do: :erlang.nif_error(:nif_not_loaded)
  def fft_frequencies(_fs, _fft_length, _endpoint), do: :erlang.nif_error(:nif_not_loaded)
  def mel_filters(_fft_length, _mel_bins, _fs, _max_mel, _spacing), do: :erlang.nif_error(:nif_not_loaded)
  def sinc(_t), do: :erlang.nif_error(:nif_not_loaded)
  def stft(_ctx, _x, _length, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def stft_c64(_ctx, _x, _length, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def istft(_ctx, _z, _frames, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def istft_filtered(_ctx, _z, _frames, _batch, _window, _params, _h), do: :erlang.nif_error(:nif_not_loaded)
  def fir(_ctx, _x, _length, _batch, _taps, _mode), do: :erlang.nif_error(:nif_not_loaded)

  def as_windowed(_ctx, _x, _length, _batch, _window_length, _stride, _pad_mode, _lo, _hi),
    do: :erlang.nif_error(:nif_not_loaded)

  def overlap_and_add(_ctx, _frames, _num_frames, _batch, _frame_length, _overlap, _components),
    do: :erlang.nif_error(:nif_not_loaded)

  def fft(_ctx, _in, _is_real, _rows, _n_in, _fft_length, _inverse), do: :erlang.nif_error(:nif_not_loaded)
  def fftconvolve_c64(_ctx, _a, _b, _mode), do: :erlang.nif_error(:nif_not_loaded)
  def fft_nd(_ctx, _in, _is_real, _shape, _axes, _lengths, _inverse), do: :erlang.nif_error(:nif_not_loaded)

  def fftconvolve_nd(_ctx, _a, _a_is_real, _a_shape, _b, _b_is_real, _b_shape, _mode),
  def convolve_direct(_ctx, _a, _a_is_real, _a_shape, _b, _b_is_real, _b_shape, _mode),
    do: :erlang.nif_error(:nif_not_loaded)
  def stft_to_mel(_ctx, _z, _rows, _fft_length, _mel_bins, _filters), do: :erlang.nif_error(:nif_not_loaded)

  def stft_mel(_ctx, _x, _length, _batch, _window, _params, _mel_bins, _filters),
    do: :erlang.nif_error(:nif_not_loaded)

  def stft_magnitude(_ctx, _x, _length, _batch, _window, _params, _kind), do: :erlang.nif_error(:nif_not_loaded)
  def to_device(_ctx, _bin), do: :erlang.nif_error(:nif_not_loaded)
  def from_device(_buf), do: :erlang.nif_error(:nif_not_loaded)
  def buf_size(_buf), do: :erlang.nif_error(:nif_not_loaded)
  def stft_dev(_ctx, _x, _length, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def stft_c64_dev(_ctx, _x, _length, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def stft_onesided_dev(_ctx, _x, _length, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def stft_packed_dev(_ctx, _x, _length, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def istft_packed_dev(_ctx, _z, _frames, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def istft_dev(_ctx, _z, _frames, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def istft_filtered_dev(_ctx, _z, _frames, _batch, _window, _params, _h), do: :erlang.nif_error(:nif_not_loaded)
  def fir_dev(_ctx, _x, _length, _batch, _taps, _mode), do: :erlang.nif_error(:nif_not_loaded)
  def spectrum_mul_dev(_ctx, _z, _rows, _fft_length, _h), do: :erlang.nif_error(:nif_not_loaded)
  def group_create(_devices), do: :erlang.nif_error(:nif_not_loaded)
  def group_info(_group), do: :erlang.nif_error(:nif_not_loaded)

  def stft_sharded(_group, _x, _length, _batch, _window, _params, _axis, _gather),
    do: :erlang.nif_error(:nif_not_loaded)

  def istft_sharded(_group, _z, _frames, _batch, _window, _params, _axis, _gather),
    do: :erlang.nif_error(:nif_not_loaded)

  def fir_sharded(_group, _x, _length, _batch, _taps, _mode, _axis, _gather),
    do: :erlang.nif_error(:nif_not_loaded)

  def stft_mel_sharded(_group, _x, _length, _batch, _window, _params, _mel_bins, _filters, _axis),
    do: :erlang.nif_error(:nif_not_loaded)

  # device-resident shards (NxSignalAMD.Sharded.Tensor): one dense buffer per group member
  def shard_range(_kind, _a, _b, _c, _world, _rank), do: :erlang.nif_error(:nif_not_loaded)
  def group_scatter(_group, _bin, _batch, _row_bytes, _parts), do: :erlang.nif_error(:nif_not_loaded)
  def group_gather(_group, _bufs, _batch, _row_bytes, _parts), do: :erlang.nif_error(:nif_not_loaded)

  def stft_sharded_dev(_group, _x_bufs, _length, _batch, _window, _params, _axis, _gather),
    do: :erlang.nif_error(:nif_not_loaded)

  def istft_sharded_dev(_group, _z_bufs, _frames, _batch, _window, _params, _axis, _gather),
    do: :erlang.nif_error(:nif_not_loaded)

  def fir_sharded_dev(_group, _x_bufs, _length, _batch, _taps, _mode, _axis, _gather),
    do: :erlang.nif_error(:nif_not_loaded)

  def stft_mel_sharded_dev(_group, _x_bufs, _length, _batch, _window, _params, _mel_bins, _filters, _axis),
    do: :erlang.nif_error(:nif_not_loaded)

  # f64 / c128 tier (include/nxsig.h): the reference computes in the type of its operands
  def window_f64(_kind, _n, _periodic, _beta, _eps), do: :erlang.nif_error(:nif_not_loaded)
  def firwin_f64(_taps, _cutoff, _kind, _beta, _pass_zero, _scale, _fs), do: :erlang.nif_error(:nif_not_loaded)
  def fft_frequencies_f64(_fs, _fft_length, _endpoint), do: :erlang.nif_error(:nif_not_loaded)
  def sinc_f64(_t), do: :erlang.nif_error(:nif_not_loaded)
  def stft_f64(_ctx, _x, _length, _batch, _window, _window_is_f64, _params), do: :erlang.nif_error(:nif_not_loaded)
  def stft_c128(_ctx, _x, _length, _batch, _window, _window_is_f64, _params), do: :erlang.nif_error(:nif_not_loaded)
  def istft_c128(_ctx, _z, _frames, _batch, _window, _window_is_f64, _params), do: :erlang.nif_error(:nif_not_loaded)
  def fir_f64(_ctx, _x, _length, _batch, _taps, _mode), do: :erlang.nif_error(:nif_not_loaded)
  def fft_c128(_ctx, _in, _is_real, _rows, _n_in, _fft_length, _inverse), do: :erlang.nif_error(:nif_not_loaded)

  def as_windowed_f64(_ctx, _x, _length, _batch, _window_length, _stride, _pad_mode, _lo, _hi),
    do: :erlang.nif_error(:nif_not_loaded)

  def overlap_and_add_f64(_ctx, _frames, _num_frames, _batch, _frame_length, _overlap, _components),
    do: :erlang.nif_error(:nif_not_loaded)
end
