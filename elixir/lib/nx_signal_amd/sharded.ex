defmodule NxSignalAMD.Sharded do
  @moduledoc """
  Multi-GPU sharding of the STFT / FIR path from ONE BEAM process (SURVEY §8e): a group holds one context + HIP stream
  per GPU and the RCCL communicators (`ncclCommInitAll`).  Channels — the reference's vectorized axes,
  `lib/nx_signal.ex:358-363` — or frame ranges of one long stream are split into contiguous blocks, every GPU computes
  its block concurrently with no data-path exchange, and the result is assembled either by per-shard downloads
  (`gather: false`, default) or by an RCCL all-gather over xGMI followed by one download (`gather: true`).
  """
  alias NxSignalAMD.NIF

  @axes %{channels: 0, frames: 1, samples: 1}
  @modes %{full: 0, same: 1, valid: 2}

  @doc "Creates a group over the given device ids (default: every GPU of the node)."
  def group(devices \\ nil) do
    devices =
      devices ||
        case NIF.device_count() |> NxSignalAMD.unwrap!() do
          {:ok, n} -> Enum.to_list(0..(n - 1))
        end

    {:ok, g} = NIF.group_create(devices) |> NxSignalAMD.unwrap!()
    g
  end

  @doc "`{world, local_count, has_rccl}`"
  def info(group), do: NIF.group_info(group)

  @doc "`NxSignal.stft/3` sharded over `group` (`window_padding: :valid` only). Options: stft's plus `axis:` and `gather:`."
  def stft(group, %Nx.Tensor{} = data, window, opts \\ []) do
    {shard_opts, stft_opts} = Keyword.split(opts, [:axis, :gather])
    axis = Map.fetch!(@axes, shard_opts[:axis] || :channels)
    gather = if shard_opts[:gather], do: 1, else: 0
    {params, fft_length} = NxSignalAMD.stft_params!(window, stft_opts)
    vec_axes = data.vectorized_axes
    flat = if vec_axes == [], do: data, else: Nx.devectorize(data, keep_names: false)
    {batch_shape, length} = NxSignalAMD.split_last(Nx.shape(flat))
    x = flat |> Nx.as_type(:f32) |> Nx.to_binary()
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, z, m} =
      NIF.stft_sharded(group, x, length, Tuple.product(batch_shape), w, params, axis, gather) |> NxSignalAMD.unwrap!()

    shape = batch_shape |> Tuple.insert_at(tuple_size(batch_shape), m) |> Tuple.insert_at(tuple_size(batch_shape) + 1, fft_length)
    z = Nx.from_binary(z, :c64) |> Nx.reshape(shape)
    if vec_axes == [], do: z, else: Nx.vectorize(z, vec_axes)
  end

  @doc """
  `NxSignal.istft/3` sharded over `group` by channels or by frame ranges (`axis: :frames`: every GPU recomputes the
  `ceil(N / hop) - 1` halo frames in front of its range instead of exchanging partial overlap-add sums). Bit-identical to
  the unsharded call. Options: istft's plus `axis:` and `gather:`.
  """
  def istft(group, %Nx.Tensor{} = data, window, opts \\ []) do
    {shard_opts, istft_opts} = Keyword.split(opts, [:axis, :gather])
    axis = Map.fetch!(@axes, shard_opts[:axis] || :channels)
    gather = if shard_opts[:gather], do: 1, else: 0
    vec_axes = data.vectorized_axes
    flat = if vec_axes == [], do: data, else: Nx.devectorize(data, keep_names: false)
    {params, _overlap, m, batch_shape} = NxSignalAMD.istft_params!(Nx.shape(flat), window, istft_opts)
    z = flat |> Nx.as_type(:c64) |> Nx.to_binary()
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, y} =
      NIF.istft_sharded(group, z, m, Tuple.product(batch_shape), w, params, axis, gather) |> NxSignalAMD.unwrap!()

    out_len = m * elem(params, 1) + (elem(params, 0) - elem(params, 1))
    y = Nx.from_binary(y, :c64) |> Nx.reshape(Tuple.insert_at(batch_shape, tuple_size(batch_shape), out_len))
    if vec_axes == [], do: y, else: Nx.vectorize(y, vec_axes)
  end

  @doc """
  `NxSignalAMD.mel_spectrogram/3` (`stft/3 |> stft_to_mel/3`, fused) sharded over `group` — one of the two sharded calls with an
  exchange step: the clamp against `reduce_max(log_spec) - 8` (`lib/nx_signal.ex:511`) takes the maximum over the WHOLE tensor,
  so the members' running maxima are all-reduced (RCCL `ncclAllReduce` / `ncclMax` on one `int32` pair) between the two passes.
  Options: mel_spectrogram's plus `axis:` (`:channels` or `:frames`); `window_padding` must be `:valid`.
  """
  def mel_spectrogram(group, %Nx.Tensor{} = data, window, opts \\ []) do
    {shard_opts, opts} = Keyword.split(opts, [:axis])
    {mel_opts, stft_opts} = Keyword.split(opts, [:mel_bins, :max_mel, :mel_frequency_spacing])
    axis = Map.fetch!(@axes, shard_opts[:axis] || :channels)
    mel_bins = mel_opts[:mel_bins] || raise ArgumentError, "missing :mel_bins option"
    {params, fft_length} = NxSignalAMD.stft_params!(window, stft_opts)
    filters = NxSignalAMD.mel_filters(fft_length, mel_bins, elem(params, 7), Keyword.delete(mel_opts, :mel_bins)) |> Nx.to_binary()
    vec_axes = data.vectorized_axes
    flat = if vec_axes == [], do: data, else: Nx.devectorize(data, keep_names: false)
    {batch_shape, length} = NxSignalAMD.split_last(Nx.shape(flat))
    x = flat |> Nx.as_type(:f32) |> Nx.to_binary()
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, out, m} =
      NIF.stft_mel_sharded(group, x, length, Tuple.product(batch_shape), w, params, mel_bins, filters, axis) |> NxSignalAMD.unwrap!()

    shape = batch_shape |> Tuple.insert_at(tuple_size(batch_shape), m) |> Tuple.insert_at(tuple_size(batch_shape) + 1, mel_bins)
    y = Nx.from_binary(out, :f32) |> Nx.reshape(shape)
    if vec_axes == [], do: y, else: Nx.vectorize(y, vec_axes)
  end

  @doc "FIR filtering (`NxSignalAMD.Filters.fir/3`) sharded over `group` by channels or by output-sample ranges."
  def fir(group, %Nx.Tensor{} = x, taps, opts \\ []) do
    opts = Keyword.validate!(opts, mode: :same, axis: :channels, gather: false)
    mode = Map.fetch!(@modes, opts[:mode])
    axis = Map.fetch!(@axes, opts[:axis])
    {batch_shape, length} = NxSignalAMD.split_last(Nx.shape(x))
    batch = Tuple.product(batch_shape)
    xb = x |> Nx.as_type(:f32) |> Nx.to_binary()
    hb = taps |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, y} =
      NIF.fir_sharded(group, xb, length, batch, hb, mode, axis, if(opts[:gather], do: 1, else: 0)) |> NxSignalAMD.unwrap!()

    n_out = div(byte_size(y), 4 * max(batch, 1))
    Nx.from_binary(y, :f32) |> Nx.reshape(Tuple.insert_at(batch_shape, tuple_size(batch_shape), n_out))
  end
end
